TAG=${1:-nahprio}
mkdir -p gpurun_out/$TAG
run() {
  local name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-to-tol --no-cpu-baseline > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err || { echo "$name FAILED"; return 0; }
  python -c "
import json; d=json.load(open('gpurun_out/$TAG/$name.json')); r=d['roofline']; print('$name', 'kernel_ms', round(r['kernel_ms_avg'],3), d['config'].get('launch'))"
}
for q in 1250 2500 5000; do for hp in 0 1 0 1; do run q${q}_hp${hp}_$RANDOM AQ_BENCH_NA=0.05 AQ_BENCH_Q=$q AQ_HPRIO=$hp; done; done
for hp in 0 1 0 1; do run c2na_hp${hp}_$RANDOM AQ_BENCH_NA=0.05 AQ_BENCH_P=5000 AQ_BENCH_Q=1000 AQ_HPRIO=$hp; done
