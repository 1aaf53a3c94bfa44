# same-box A/B of two builds at C3 (interleaved, 3 rounds): bash tools/gpu_ab_c3.sh <tag> <libA> <libB>
TAG=$1; A=$2; B=$3
mkdir -p gpurun_out/$TAG
for rep in 1 2 3; do for L in A B; do
  lib=$A; [ $L = B ] && lib=$B
  AQ_LIB=$PWD/$lib timeout -k 10 200 python bench.py --no-to-tol --no-cpu-baseline > gpurun_out/$TAG/c3_${L}_$rep.json 2> gpurun_out/$TAG/c3_${L}_$rep.err || { echo FAILED $L; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/$TAG/c3_${L}_$rep.json')); r=d['roofline']; print('c3', '$L', $rep, 'kernel_ms', round(r['kernel_ms_avg'],3), 'frac', round(r['frac'],4), 'sweeps/s', round(d['value'],2))"
done; done
