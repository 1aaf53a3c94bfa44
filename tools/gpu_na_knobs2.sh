TAG=${1:-naknobs2}
mkdir -p gpurun_out/$TAG
run() {
  local name=$1; shift
  env AQ_BENCH_NA=0.05 "$@" timeout -k 10 200 python bench.py --no-to-tol --no-cpu-baseline > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err || { echo "$name FAILED"; return 1; }
  python -c "
import json; d=json.load(open('gpurun_out/$TAG/$name.json')); r=d['roofline']; print('$name', 'kernel_ms', round(r['kernel_ms_avg'],3), 'ms/step', round(d['ms_per_step'],3))"
}
run base AQ_DUMMY=1 && run hprio1 AQ_HPRIO=1 && run hprio2 AQ_HPRIO=2 && run base_b AQ_DUMMY=1 && run hprio1_b AQ_HPRIO=1 && run hprio2_b AQ_HPRIO=2 && run hprio1_mprio2 AQ_HPRIO=1 AQ_MPRIO=0
