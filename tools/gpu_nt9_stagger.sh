TAG=${1:-nt9stagger}
mkdir -p gpurun_out/$TAG
run() {
  local name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-to-tol --no-cpu-baseline > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err || { echo "$name FAILED"; tail -2 gpurun_out/$TAG/$name.err; return 0; }
  python -c "
import json; d=json.load(open('gpurun_out/$TAG/$name.json')); r=d['roofline']; print('$name', 'kernel_ms', round(r['kernel_ms_avg'],3), 'frac', round(r['frac'],4), 'sweeps/s', round(d['value'],2), d['config'].get('launch'), d['config'].get('env_overrides'))"
}
run default AQ_DUMMY=1
for st in 0 1 2 3 4 6; do run st$st AQ_STAGGER=$st; done
run nt6_st0 AQ_NT3=6 AQ_STAGGER=0
run default_b AQ_DUMMY=1
run st2_b AQ_STAGGER=2
run st0_b AQ_STAGGER=0
run q8000 AQ_BENCH_Q=8000
run q8000_nt6 AQ_BENCH_Q=8000 AQ_NT3=6 AQ_STAGGER=0
