TAG=${1:-chaincheck}
mkdir -p gpurun_out/$TAG
run() {
  local name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-to-tol --no-cpu-baseline > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err || { echo "$name FAILED"; return 0; }
  python -c "
import json; d=json.load(open('gpurun_out/$TAG/$name.json')); r=d['roofline']; print('$name', 'kernel_ms', round(r['kernel_ms_avg'],3), 'sweeps/s', round(d['value'],2), d['config'].get('launch'))"
}
run c3 AQ_DUMMY=1; run c3_chain4 AQ_CHAIN=4; run q5000 AQ_BENCH_Q=5000; run q5000_chain4 AQ_BENCH_Q=5000 AQ_CHAIN=4; run na AQ_BENCH_NA=0.05; run q7000 AQ_BENCH_Q=7000; run q7000_chain4 AQ_BENCH_Q=7000 AQ_CHAIN=4; run c3_b AQ_DUMMY=1; run c3_chain4_b AQ_CHAIN=4
