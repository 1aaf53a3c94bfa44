TAG=${1:-c5hprio}
mkdir -p gpurun_out/$TAG
run() {
  local name=$1; shift
  env AQ_BENCH_N=5000 AQ_BENCH_P=200000 AQ_BENCH_Q=2500 AQ_BENCH_NA=0.05 "$@" timeout -k 10 400 python bench.py --steps 5 --warmup 10 --no-to-tol --no-cpu-baseline > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err || { echo "$name FAILED"; return 1; }
  python -c "
import json; d=json.load(open('gpurun_out/$TAG/$name.json')); r=d['roofline']; print('$name', 'kernel_ms', round(r['kernel_ms_avg'],3), 'ms/step', round(d['ms_per_step'],3))"
}
run base AQ_DUMMY=1 && run hprio1 AQ_HPRIO=1 && run base_b AQ_DUMMY=1 && run hprio1_b AQ_HPRIO=1
