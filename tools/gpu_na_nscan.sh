# where does a MASK phase go: time per SNP block against n (the matrix waves' share shrinks with n, the chain / helper cycle does not)
TAG=${1:-nanscan}
mkdir -p gpurun_out/$TAG
run() {
  local name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-to-tol --no-cpu-baseline > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err || { echo "$name FAILED"; return 0; }
  python -c "
import json; d=json.load(open('gpurun_out/$TAG/$name.json')); r=d['roofline']; c=d['config']; nb=(c['p']+15)//16; print('$name', 'kernel_ms', round(r['kernel_ms_avg'],3), 'us/block', round(r['kernel_ms_avg']*1e3/nb,3), c.get('launch'))"
}
for n in 1000 750 500 250 125; do
  run na_n$n AQ_BENCH_NA=0.05 AQ_BENCH_Q=1250 AQ_BENCH_N=$n
  run cy_n$n AQ_BENCH_Q=1250 AQ_BENCH_N=$n AQ_LA_NOSPLIT=1
done
