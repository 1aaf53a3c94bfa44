# chain-bound shapes: sample split C x who exchanges (AQ_LA_XHELPER) x helper priority; bash tools/gpu_split_knobs.sh <tag>
TAG=${1:-splitknobs}
mkdir -p gpurun_out/$TAG
run() {
  local name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-to-tol --no-cpu-baseline > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err || { echo "$name FAILED"; tail -2 gpurun_out/$TAG/$name.err; return 0; }
  python -c "
import json; d=json.load(open('gpurun_out/$TAG/$name.json')); r=d['roofline']; print('$name', 'kernel_ms', round(r['kernel_ms_avg'],3), d['config'].get('launch'))"
}
for C in 2 3; do for XH in 0 1; do for HP in 0 1; do
  run q1250_C${C}_xh${XH}_hp${HP} AQ_BENCH_Q=1250 AQ_LA_C=$C AQ_LA_XHELPER=$XH AQ_HPRIO=$HP
done; done; done
for C in 2 3 4; do for XH in 0 1; do
  run c2_C${C}_xh${XH} AQ_BENCH_P=5000 AQ_BENCH_Q=1000 AQ_LA_C=$C AQ_LA_XHELPER=$XH
done; done
