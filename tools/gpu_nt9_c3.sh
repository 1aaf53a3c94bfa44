TAG=${1:-nt9c3}
mkdir -p gpurun_out/$TAG
run() {
  local name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-to-tol --no-cpu-baseline > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err || { echo "$name FAILED"; return 0; }
  python -c "
import json; d=json.load(open('gpurun_out/$TAG/$name.json')); r=d['roofline']; print('$name', 'kernel_ms', round(r['kernel_ms_avg'],3), 'frac', round(r['frac'],4), 'sweeps/s', round(d['value'],2), 'annealed', round(d['config']['annealed_sweeps_per_s'],2))"
}
run base_st0 AQ_STAGGER=0
for hp in 1 2 3; do for mp in 1 0; do run nt9_hp${hp}_mp${mp} AQ_STAGGER=0 AQ_NT3=9 AQ_HPRIO=$hp AQ_MPRIO=$mp; done; done
run nt9_hp1_st2 AQ_STAGGER=2 AQ_NT3=9 AQ_HPRIO=1
run nt9_hp1_xt0 AQ_STAGGER=0 AQ_NT3=9 AQ_HPRIO=1 AQ_XTOUCH=0
run base_st0_b AQ_STAGGER=0
run nt9_hp1_b AQ_STAGGER=0 AQ_NT3=9 AQ_HPRIO=1
