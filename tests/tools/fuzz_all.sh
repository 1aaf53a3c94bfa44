# Randomised parity sweep (tests/tools/fuzz_parity.py) under the launch forms the host would not pick at these sizes: one process
# per form, ONE AFTER THE OTHER (a fault under five concurrent processes could not be traced to one launch form).
# usage (GPU box): bash tests/tools/fuzz_all.sh <ncases> <seed>   -> gpurun_out/fuzz_*.log
N=${1:-100}
SEED=${2:-777}
mkdir -p gpurun_out
rc=0
python tests/tools/fuzz_parity.py $N $SEED > gpurun_out/fuzz_default.log 2>&1 || rc=1
AQ_TT=2 python tests/tools/fuzz_parity.py $N $((SEED+1)) > gpurun_out/fuzz_tt2.log 2>&1 || rc=1
AQ_LA_C=2 python tests/tools/fuzz_parity.py $N $((SEED+2)) > gpurun_out/fuzz_split2.log 2>&1 || rc=1
AQ_LA_C=3 AQ_LA_XHELPER=1 python tests/tools/fuzz_parity.py $N $((SEED+3)) > gpurun_out/fuzz_split3_helper.log 2>&1 || rc=1
AQ_CHAIN=3 python tests/tools/fuzz_parity.py $N $((SEED+4)) > gpurun_out/fuzz_chain3.log 2>&1 || rc=1
AQ_TT=1 AQ_NT3=6 AQ_LA_NOSPLIT=1 python tests/tools/fuzz_parity.py $N $((SEED+5)) > gpurun_out/fuzz_tt1_nt3_6.log 2>&1 || rc=1
AQ_TT=2 AQ_NT3=9 AQ_CHAIN=2 python tests/tools/fuzz_parity.py $N $((SEED+6)) > gpurun_out/fuzz_tt2_nt3_9.log 2>&1 || rc=1
tail -n 1 gpurun_out/fuzz_*.log
grep -l "MISMATCH\|Traceback" gpurun_out/fuzz_*.log && exit 1
exit $rc
