# Quick measurement pass on the GPU box (one gpurun call): parity subset, default bench, role counters + timeline of the diag build.
# usage: bash tools/gpu_quick.sh <tag>  -> gpurun_out/<tag>/
TAG=${1:-quick}
mkdir -p gpurun_out/$TAG
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_special.py tests/test_gpu_bigp.py tests/test_gpu_sharded.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/$TAG/pytest.log 2>&1
echo rc=$? >> gpurun_out/$TAG/pytest.log
tail -4 gpurun_out/$TAG/pytest.log
timeout -k 10 300 python bench.py --no-to-tol --no-cpu-baseline > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
python -c "
import json; d=json.load(open('gpurun_out/$TAG/bench.json')); print('sweeps/s', d['value'], 'ms/step', d['ms_per_step'], 'kernel ms', d['roofline']['kernel_ms_avg'], 'frac', d['roofline']['frac'], 'annealed sweeps/s', d['config']['annealed_sweeps_per_s'])"
AQ_LIB=$PWD/atlasqtl_amd/libatlasqtl_hip_diag.so AQ_DIAG_DUMP=$PWD/gpurun_out/$TAG/roles.txt timeout -k 10 300 python bench.py --steps 10 --warmup 10 --no-cpu-baseline --no-to-tol > gpurun_out/$TAG/bench_diag.json 2> gpurun_out/$TAG/bench_diag.err
python tools/prof_roles.py gpurun_out/$TAG/roles.txt > gpurun_out/$TAG/roles_summary.txt
python tools/prof_timeline.py gpurun_out/$TAG/roles.txt.timeline > gpurun_out/$TAG/timeline.txt
cat gpurun_out/$TAG/roles_summary.txt
grep -E "^ +3 " gpurun_out/$TAG/timeline.txt | head -4
tail -1 gpurun_out/$TAG/timeline.txt
