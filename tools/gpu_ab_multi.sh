# same-box A/B of two builds on C3, C3 + NA, q = 5000, q = 1250 (interleaved, 2 rounds): bash tools/gpu_ab_multi.sh <tag> <libA> <libB>
TAG=$1; A=$2; B=$3
mkdir -p gpurun_out/$TAG
for rep in 1 2; do for L in A B; do
  lib=$A; [ $L = B ] && lib=$B
  for cfg in c3 na q5000 q1250; do
    case $cfg in c3) env="AQ_DUMMY=1";; na) env="AQ_BENCH_NA=0.05";; q5000) env="AQ_BENCH_Q=5000";; q1250) env="AQ_BENCH_Q=1250";; esac
    env $env AQ_LIB=$PWD/$lib timeout -k 10 200 python bench.py --no-to-tol --no-cpu-baseline > gpurun_out/$TAG/${cfg}_${L}_$rep.json 2> gpurun_out/$TAG/${cfg}_${L}_$rep.err || { echo FAILED $cfg $L; exit 1; }
    python -c "
import json; d=json.load(open('gpurun_out/$TAG/${cfg}_${L}_$rep.json')); r=d['roofline']; print('$cfg', '$L', $rep, 'kernel_ms', round(r['kernel_ms_avg'],3))"
  done
done; done
