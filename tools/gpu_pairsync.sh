# the two matrix waves of a SIMD enter every phase together (AQ_PAIRSYNC=1) against the default; bash tools/gpu_pairsync.sh <tag>
TAG=${1:-pairsync}
mkdir -p gpurun_out/$TAG
run() {
  local name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-to-tol --no-cpu-baseline > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err || { echo "$name FAILED"; tail -2 gpurun_out/$TAG/$name.err; return 0; }
  python -c "
import json; d=json.load(open('gpurun_out/$TAG/$name.json')); r=d['roofline']; print('$name', 'kernel_ms', round(r['kernel_ms_avg'],3))"
}
for rep in 1 2; do
run na_base_$rep AQ_BENCH_NA=0.05
run na_ps_$rep AQ_BENCH_NA=0.05 AQ_PAIRSYNC=1
run c3_base_$rep AQ_DUMMY=1
run c3_ps_st0_$rep AQ_PAIRSYNC=1 AQ_STAGGER=0
run q5000_base_$rep AQ_BENCH_Q=5000
run q5000_ps_$rep AQ_BENCH_Q=5000 AQ_PAIRSYNC=1
run q1250_base_$rep AQ_BENCH_Q=1250
run q1250_ps_$rep AQ_BENCH_Q=1250 AQ_PAIRSYNC=1
done
