TAG=${1:-stagger2c3}
mkdir -p gpurun_out/$TAG
run() {
  local name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-to-tol --no-cpu-baseline > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err || { echo "$name FAILED"; return 0; }
  python -c "
import json; d=json.load(open('gpurun_out/$TAG/$name.json')); r=d['roofline']; print('$name', 'kernel_ms', round(r['kernel_ms_avg'],3), 'frac', round(r['frac'],4))"
}
run st0 AQ_STAGGER=0
run st0_mprio0 AQ_STAGGER=0 AQ_MPRIO=0
run st0_hprio1 AQ_STAGGER=0 AQ_HPRIO=1
run st0_nt3_9 AQ_STAGGER=0 AQ_NT3=9
run st0_nt3_9_hprio1 AQ_STAGGER=0 AQ_NT3=9 AQ_HPRIO=1
run st0_xtouch0 AQ_STAGGER=0 AQ_XTOUCH=0
run st0_b AQ_STAGGER=0
run st0_na AQ_STAGGER=0 AQ_BENCH_NA=0.05
