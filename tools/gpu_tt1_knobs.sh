TAG=${1:-tt1knobs}
mkdir -p gpurun_out/$TAG
run() {
  local name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-to-tol --no-cpu-baseline > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err || { echo "$name FAILED"; return 0; }
  python -c "
import json; d=json.load(open('gpurun_out/$TAG/$name.json')); r=d['roofline']; print('$name', 'kernel_ms', round(r['kernel_ms_avg'],3))"
}
for q in 5000 2500; do
  run q${q}_base AQ_BENCH_Q=$q
  run q${q}_st3 AQ_BENCH_Q=$q AQ_STAGGER=3
  run q${q}_st5 AQ_BENCH_Q=$q AQ_STAGGER=5
  run q${q}_mp0 AQ_BENCH_Q=$q AQ_MPRIO=0
  run q${q}_st3_hp1 AQ_BENCH_Q=$q AQ_STAGGER=3 AQ_HPRIO=1
  run q${q}_nt0 AQ_BENCH_Q=$q AQ_NT3=0
  run q${q}_base_b AQ_BENCH_Q=$q
done
