# helper-wave priority x recurrence-wave tile count at C3, and the N = 2 / 8 shards; usage: bash tools/gpu_hprio.sh <tag>
TAG=${1:-hprio}
mkdir -p gpurun_out/$TAG
for nt3 in 6 9; do for hp in 0 1 3; do
  AQ_NT3=$nt3 AQ_HPRIO=$hp timeout -k 10 200 python bench.py --no-to-tol --no-cpu-baseline > gpurun_out/$TAG/b_${nt3}_$hp.json 2> gpurun_out/$TAG/b_${nt3}_$hp.err
  python -c "
import json; d=json.load(open('gpurun_out/$TAG/b_${nt3}_$hp.json')); r=d['roofline']; print('C3 NT3', $nt3, 'HPRIO', $hp, 'kernel_ms', round(r['kernel_ms_avg'],3), 'frac', round(r['frac'],4), 'annealed', round(d['config']['annealed_sweeps_per_s'],2))"
done; done
for q in 5000 1250; do for hp in 0 3; do
  AQ_HPRIO=$hp AQ_BENCH_Q=$q timeout -k 10 200 python bench.py --no-to-tol --no-cpu-baseline > gpurun_out/$TAG/bq_${q}_$hp.json 2> gpurun_out/$TAG/bq_${q}_$hp.err
  python -c "
import json; d=json.load(open('gpurun_out/$TAG/bq_${q}_$hp.json')); r=d['roofline']; print('q', $q, 'HPRIO', $hp, 'kernel_ms', round(r['kernel_ms_avg'],3))"
done; done
AQ_BENCH_NA=0.05 AQ_HPRIO=3 timeout -k 10 200 python bench.py --no-to-tol --no-cpu-baseline > gpurun_out/$TAG/b_na_hp3.json 2> gpurun_out/$TAG/b_na_hp3.err
python -c "
import json; d=json.load(open('gpurun_out/$TAG/b_na_hp3.json')); r=d['roofline']; print('C3+NA HPRIO 3 kernel_ms', round(r['kernel_ms_avg'],3))"
