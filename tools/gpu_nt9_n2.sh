TAG=${1:-nt9n2}
mkdir -p gpurun_out/$TAG
run() {
  local name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-to-tol --no-cpu-baseline > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err || { echo "$name FAILED"; tail -2 gpurun_out/$TAG/$name.err; return 0; }
  python -c "
import json; d=json.load(open('gpurun_out/$TAG/$name.json')); r=d['roofline']; print('$name', 'kernel_ms', round(r['kernel_ms_avg'],3), 'us/phase', round(r['kernel_ms_avg']/3.888,3))"
}
for n in 800 900; do for nt in 3 6; do for hp in 0 1; do run n${n}_nt${nt}_hp${hp} AQ_BENCH_N=$n AQ_NT3=$nt AQ_HPRIO=$hp; done; done; done
run n1000_nt3_hp0 AQ_NT3=3 AQ_HPRIO=0
run n1000_nt3_hp1 AQ_NT3=3 AQ_HPRIO=1
run n700_nt3_hp0 AQ_BENCH_N=700 AQ_NT3=3 AQ_HPRIO=0
run n700_default AQ_BENCH_N=700
run n960_default AQ_BENCH_N=960
run n960_nt6 AQ_BENCH_N=960 AQ_NT3=6 AQ_HPRIO=0
run n960_nt3 AQ_BENCH_N=960 AQ_NT3=3 AQ_HPRIO=0
