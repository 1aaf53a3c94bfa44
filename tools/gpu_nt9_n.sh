TAG=${1:-nt9n}
mkdir -p gpurun_out/$TAG
run() {
  local name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-to-tol --no-cpu-baseline > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err || { echo "$name FAILED"; tail -2 gpurun_out/$TAG/$name.err; return 0; }
  python -c "
import json; d=json.load(open('gpurun_out/$TAG/$name.json')); r=d['roofline']; print('$name', 'kernel_ms', round(r['kernel_ms_avg'],3), 'TFLOP/s', round(r['achieved'],2))"
}
for n in 900 1056 800; do
  run n${n}_default AQ_BENCH_N=$n
  run n${n}_nt6 AQ_BENCH_N=$n AQ_NT3=6 AQ_HPRIO=0
  run n${n}_nt3 AQ_BENCH_N=$n AQ_NT3=3 AQ_HPRIO=0
done
