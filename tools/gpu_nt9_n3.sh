TAG=${1:-nt9n3}
mkdir -p gpurun_out/$TAG
run() {
  local name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-to-tol --no-cpu-baseline > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err || { echo "$name FAILED"; tail -2 gpurun_out/$TAG/$name.err; return 0; }
  python -c "
import json; d=json.load(open('gpurun_out/$TAG/$name.json')); r=d['roofline']; print('$name', 'kernel_ms', round(r['kernel_ms_avg'],3), 'us/phase', round(r['kernel_ms_avg']/3.888,3), 'TFLOP/s', round(r['achieved'],2))"
}
for n in 600 700 800 900 960 1000 1056; do run n${n}_default AQ_BENCH_N=$n; done
