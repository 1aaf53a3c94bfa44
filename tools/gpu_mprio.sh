# matrix-wave priority switching (AQ_MPRIO) at C3, shards, C3 + NA; usage: bash tools/gpu_mprio.sh <tag>
TAG=${1:-mprio}
mkdir -p gpurun_out/$TAG
for mp in 0 1; do for st in "" 0; do
  AQ_MPRIO=$mp ${st:+AQ_STAGGER=$st} timeout -k 10 200 python bench.py --no-to-tol --no-cpu-baseline > gpurun_out/$TAG/b_${mp}_s$st.json 2> gpurun_out/$TAG/b_${mp}_s$st.err
  python -c "
import json; d=json.load(open('gpurun_out/$TAG/b_${mp}_s$st.json')); r=d['roofline']; print('C3 MPRIO', $mp, 'stagger', '$st' or 'default', 'kernel_ms', round(r['kernel_ms_avg'],3), 'frac', round(r['frac'],4), 'annealed', round(d['config']['annealed_sweeps_per_s'],2))"
done; done
for q in 5000 2500 1250; do for mp in 0 1; do
  AQ_MPRIO=$mp AQ_BENCH_Q=$q timeout -k 10 200 python bench.py --no-to-tol --no-cpu-baseline > gpurun_out/$TAG/bq_${q}_$mp.json 2> gpurun_out/$TAG/bq_${q}_$mp.err
  python -c "
import json; d=json.load(open('gpurun_out/$TAG/bq_${q}_$mp.json')); r=d['roofline']; print('q', $q, 'MPRIO', $mp, 'kernel_ms', round(r['kernel_ms_avg'],3))"
done; done
for mp in 0 1; do
AQ_BENCH_NA=0.05 AQ_MPRIO=$mp timeout -k 10 200 python bench.py --no-to-tol --no-cpu-baseline > gpurun_out/$TAG/b_na_$mp.json 2> gpurun_out/$TAG/b_na_$mp.err
python -c "
import json; d=json.load(open('gpurun_out/$TAG/b_na_$mp.json')); r=d['roofline']; print('C3+NA MPRIO $mp kernel_ms', round(r['kernel_ms_avg'],3))"
done
