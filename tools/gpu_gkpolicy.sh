# cache policy of the Gram-block LDS-DMA stream (AQ_GLDS_AUX at build time: 2 = nt shipped, 0 none, 16 sc1, 18 sc1 nt); NA workload, one box
TAG=${1:-gkpolicy}
mkdir -p gpurun_out/$TAG
for rep in 1 2; do for v in ship g0 g16 g18; do
  lib=atlasqtl_amd/libatlasqtl_hip.so; [ $v != ship ] && lib=atlasqtl_amd/libatlasqtl_hip_$v.so
  AQ_BENCH_NA=0.05 AQ_LIB=$PWD/$lib timeout -k 10 200 python bench.py --no-to-tol --no-cpu-baseline > gpurun_out/$TAG/na_${v}_$rep.json 2> gpurun_out/$TAG/na_${v}_$rep.err || { echo FAILED $v; continue; }
  python -c "
import json; d=json.load(open('gpurun_out/$TAG/na_${v}_$rep.json')); r=d['roofline']; print('na', '$v', $rep, 'kernel_ms', round(r['kernel_ms_avg'],3))"
done; done
