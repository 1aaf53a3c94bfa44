# env-only knobs on the NA workload (C3 + 5 % NA), one box: bash tools/gpu_na_knobs.sh <tag>
TAG=${1:-naknobs}
mkdir -p gpurun_out/$TAG
run() {   # run <name> <env...>
  local name=$1; shift
  env AQ_BENCH_NA=0.05 "$@" timeout -k 10 200 python bench.py --no-to-tol --no-cpu-baseline > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err || { echo "$name FAILED"; return 1; }
  python -c "
import json; d=json.load(open('gpurun_out/$TAG/$name.json')); r=d['roofline']; print('$name', 'kernel_ms', round(r['kernel_ms_avg'],3), 'ms/step', round(d['ms_per_step'],3))"
}
run base AQ_DUMMY=1 && run xtouch0 AQ_XTOUCH=0 && run chain5 AQ_CHAIN=5 && run chain13 AQ_CHAIN=13 && run stag3 AQ_STAGGER=3 && run stag6 AQ_STAGGER=6 && run hprio1 AQ_HPRIO=1 && run mprio0 AQ_MPRIO=0 && run base2 AQ_DUMMY=1
