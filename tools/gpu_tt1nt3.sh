# one-tile workgroups with residual tiles on the recurrence wave: shards q = 5000 / 2500 at AQ_NT3 = 0 (default) / 3 / 6 / 9
TAG=${1:-tt1nt3}
mkdir -p gpurun_out/$TAG
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "recurrence_wave_tile_counts" > gpurun_out/$TAG/pytest.log 2>&1
echo rc=$? >> gpurun_out/$TAG/pytest.log; tail -3 gpurun_out/$TAG/pytest.log
for q in 5000 2500; do for v in 0 3 6 9; do
  AQ_NT3=$v AQ_BENCH_Q=$q timeout -k 10 200 python bench.py --no-to-tol --no-cpu-baseline > gpurun_out/$TAG/b_${q}_$v.json 2> gpurun_out/$TAG/b_${q}_$v.err
  python -c "
import json; d=json.load(open('gpurun_out/$TAG/b_${q}_$v.json')); r=d['roofline']; print('q', $q, 'NT3', $v, 'kernel_ms', round(r['kernel_ms_avg'],3), 'sweeps/s', round(d['value'],2), d['config']['launch'])"
done; done
