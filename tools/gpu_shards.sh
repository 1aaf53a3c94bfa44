# One-GPU model of the strong-scaling shards of C3 (q / N traits on one MI355X, N = 1, 2, 4, 8) plus C2; usage: bash tools/gpu_shards.sh <tag>
# NOT a scaling measurement: every line is one GPU running the shard a rank of an N-GPU job would hold.
TAG=${1:-shards}
mkdir -p gpurun_out/$TAG
: > gpurun_out/$TAG/shards.txt
for q in 10000 5000 2500 1250; do
  AQ_BENCH_Q=$q timeout -k 10 300 python bench.py --no-to-tol --no-cpu-baseline > gpurun_out/$TAG/b_$q.json 2> gpurun_out/$TAG/b_$q.err
  python -c "
import json; d=json.load(open('gpurun_out/$TAG/b_$q.json')); r=d['roofline']; print('q_per_gpu', $q, 'sweeps/s', round(d['value'],2), 'ms_per_step', round(d['ms_per_step'],3), 'kernel_ms', round(r['kernel_ms_avg'],3), 'TFLOP/s', round(r['achieved'],2), 'launch', d['config']['launch'])" | tee -a gpurun_out/$TAG/shards.txt
done
AQ_BENCH_P=5000 AQ_BENCH_Q=1000 timeout -k 10 300 python bench.py --no-to-tol --no-cpu-baseline --steps 50 > gpurun_out/$TAG/b_c2.json 2> gpurun_out/$TAG/b_c2.err
python -c "
import json; d=json.load(open('gpurun_out/$TAG/b_c2.json')); r=d['roofline']; print('C2 n=1000 p=5000 q=1000 sweeps/s', round(d['value'],2), 'ms_per_step', round(d['ms_per_step'],3), 'kernel_ms', round(r['kernel_ms_avg'],3), 'TFLOP/s', round(r['achieved'],2), 'launch', d['config']['launch'])" | tee -a gpurun_out/$TAG/shards.txt
