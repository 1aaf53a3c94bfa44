# Profile pass of the default bench workload (C3, 1 x MI355X): kernel stats, then FETCH_SIZE and WRITE_SIZE in separate --pmc runs
# (MI355X_MICROARCH.md, HBM section).  usage (on the GPU box): bash tools/prof_bench.sh <tag>   -> gpurun_out/<tag>/
set -e
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/stats -- python bench.py --steps 10 --warmup 10 --no-cpu-baseline --no-to-tol > gpurun_out/$TAG/stats.log 2>&1
echo stats done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/$TAG/fetch -- python bench.py --steps 2 --warmup 10 --no-cpu-baseline --no-to-tol > gpurun_out/$TAG/fetch.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/$TAG/write -- python bench.py --steps 2 --warmup 10 --no-cpu-baseline --no-to-tol > gpurun_out/$TAG/write.log 2>&1
echo write done
python tools/pmc_summary.py FETCH_SIZE=gpurun_out/$TAG/fetch WRITE_SIZE=gpurun_out/$TAG/write > gpurun_out/$TAG/pmc_summary.txt
cp $(find gpurun_out/$TAG/stats -name "*kernel_stats.csv" | head -1) gpurun_out/$TAG/kernel_stats.csv
find gpurun_out/$TAG -name "*kernel_trace.csv" -delete
find gpurun_out/$TAG -name "*counter_collection.csv" -delete
grep '"metric"' gpurun_out/$TAG/stats.log | cut -c1-200
