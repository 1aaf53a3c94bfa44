# one-tile workgroups (the N = 2 / 4 shards of C3): residual tiles on the recurrence wave (AQ_NT3) x helper priority (AQ_HPRIO)
TAG=${1:-tt1hprio}
mkdir -p gpurun_out/$TAG
for q in 5000 2500; do for nt3 in 3 6; do for hp in 0 1 2; do
  AQ_NT3=$nt3 AQ_HPRIO=$hp AQ_BENCH_Q=$q timeout -k 10 200 python bench.py --no-to-tol --no-cpu-baseline > gpurun_out/$TAG/b_${q}_${nt3}_$hp.json 2> gpurun_out/$TAG/b_${q}_${nt3}_$hp.err || { echo FAILED $q $nt3 $hp; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/$TAG/b_${q}_${nt3}_$hp.json')); r=d['roofline']; print('q', $q, 'NT3', $nt3, 'HPRIO', $hp, 'kernel_ms', round(r['kernel_ms_avg'],3))"
done; done; done
