set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01v6
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01v6/stats -- python bench.py --steps 10 --warmup 10 --no-cpu-baseline > gpurun_out/r01v6/stats.log 2>&1
echo stats done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r01v6/fetch -- python bench.py --steps 2 --warmup 10 --no-cpu-baseline > gpurun_out/r01v6/fetch.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r01v6/write -- python bench.py --steps 2 --warmup 10 --no-cpu-baseline > gpurun_out/r01v6/write.log 2>&1
echo write done
python tools/pmc_summary.py FETCH_SIZE=gpurun_out/r01v6/fetch WRITE_SIZE=gpurun_out/r01v6/write > gpurun_out/r01v6/pmc_summary.txt
find gpurun_out/r01v6 -name "*kernel_trace.csv" -delete
find gpurun_out/r01v6 -name "*counter_collection.csv" -delete
grep '"metric"' gpurun_out/r01v6/stats.log | cut -c1-200
