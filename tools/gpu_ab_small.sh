# same-box A/B of two builds on the chain-bound shapes: q = 1250 shard (N = 8) and C2; bash tools/gpu_ab_small.sh <tag> <libA> <libB>
TAG=$1; A=$2; B=$3
mkdir -p gpurun_out/$TAG
for rep in 1 2; do for L in A B; do
  lib=$A; [ $L = B ] && lib=$B
  for cfg in q1250 c2 q2500; do
    case $cfg in q1250) env="AQ_BENCH_Q=1250";; c2) env="AQ_BENCH_P=5000 AQ_BENCH_Q=1000";; q2500) env="AQ_BENCH_Q=2500";; esac
    env $env AQ_LIB=$PWD/$lib timeout -k 10 200 python bench.py --no-to-tol --no-cpu-baseline > gpurun_out/$TAG/${cfg}_${L}_$rep.json 2> gpurun_out/$TAG/${cfg}_${L}_$rep.err || { echo FAILED $cfg $L; exit 1; }
    python -c "
import json; d=json.load(open('gpurun_out/$TAG/${cfg}_${L}_$rep.json')); r=d['roofline']; print('$cfg', '$L', $rep, 'kernel_ms', round(r['kernel_ms_avg'],3), 'ms/step', round(d['ms_per_step'],3))"
  done
done; done
