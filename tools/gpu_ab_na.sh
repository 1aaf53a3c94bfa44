# same-box A/B of two builds of the library on the NA workload (C3 + 5 % NA), interleaved: bash tools/gpu_ab_na.sh <tag> <libA> <libB> [reps]
TAG=$1; A=$2; B=$3; REPS=${4:-2}
mkdir -p gpurun_out/$TAG
for rep in $(seq 1 $REPS); do for L in A B; do
  lib=$A; [ $L = B ] && lib=$B
  AQ_BENCH_NA=0.05 AQ_LIB=$PWD/$lib timeout -k 10 200 python bench.py --no-to-tol --no-cpu-baseline > gpurun_out/$TAG/na_${L}_$rep.json 2> gpurun_out/$TAG/na_${L}_$rep.err || exit 1
  python -c "
import json; d=json.load(open('gpurun_out/$TAG/na_${L}_$rep.json')); r=d['roofline']; print('na', '$L', $rep, 'kernel_ms', round(r['kernel_ms_avg'],3), 'ms/step', round(d['ms_per_step'],3), 'annealed', round(d['config']['annealed_sweeps_per_s'],2))"
done; done
