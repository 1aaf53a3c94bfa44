TAG=${1:-staggerc3}
mkdir -p gpurun_out/$TAG
for st in 4 0 1 2 6 9 4 0; do
  AQ_STAGGER=$st timeout -k 10 200 python bench.py --no-to-tol --no-cpu-baseline > gpurun_out/$TAG/c3_$st.json 2> gpurun_out/$TAG/c3_$st.err || { echo FAILED $st; continue; }
  python -c "
import json; d=json.load(open('gpurun_out/$TAG/c3_$st.json')); r=d['roofline']; print('AQ_STAGGER', $st, 'kernel_ms', round(r['kernel_ms_avg'],3), 'frac', round(r['frac'],4))"
done
