# bench variants of the recurrence wave's tile count at C3 (one gpurun call); usage: bash tools/gpu_nt3.sh <tag>
TAG=${1:-nt3}
mkdir -p gpurun_out/$TAG
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bigp.py -m gpu -x -q > gpurun_out/$TAG/pytest.log 2>&1
echo rc=$? >> gpurun_out/$TAG/pytest.log
tail -3 gpurun_out/$TAG/pytest.log
for v in 9 6 3; do
  AQ_NT3=$v timeout -k 10 300 python bench.py --no-to-tol --no-cpu-baseline > gpurun_out/$TAG/bench_nt3_$v.json 2> gpurun_out/$TAG/bench_nt3_$v.err
  python -c "
import json; d=json.load(open('gpurun_out/$TAG/bench_nt3_$v.json')); print('NT3=$v sweeps/s', d['value'], 'ms/step', d['ms_per_step'], 'kernel ms', d['roofline']['kernel_ms_avg'], 'frac', d['roofline']['frac'], 'annealed sweeps/s', d['config']['annealed_sweeps_per_s'])"
done
