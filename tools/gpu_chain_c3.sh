# C3: number of chained SNP segments per trait group (AQ_CHAIN); bash tools/gpu_chain_c3.sh <tag>
TAG=${1:-chainc3}
mkdir -p gpurun_out/$TAG
for ch in 4 13 8 16 26 4 13; do
  AQ_CHAIN=$ch timeout -k 10 200 python bench.py --no-to-tol --no-cpu-baseline > gpurun_out/$TAG/c3_$ch.json 2> gpurun_out/$TAG/c3_$ch.err || { echo FAILED $ch; continue; }
  python -c "
import json; d=json.load(open('gpurun_out/$TAG/c3_$ch.json')); r=d['roofline']; print('AQ_CHAIN', $ch, 'kernel_ms', round(r['kernel_ms_avg'],3), 'frac', round(r['frac'],4), d['config'].get('launch'))"
done
