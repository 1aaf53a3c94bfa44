# Instruction-mix counters of the bench workload (C3, 1 x MI355X): how many cycles of each SIMD's vector pipe the look-ahead kernel
# fills with MFMAs and with other VALU operations (tools/ubench_f64_pipe.hip: the two do not overlap on gfx950).
# usage (on the GPU box): bash tools/prof_instmix.sh <tag>   -> gpurun_out/<tag>/instmix_*.txt   (one --pmc pass per counter group,
# --kernel-trace only, as MI355X_MICROARCH.md prescribes)
set -e
TAG=${1:-r03mix}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
rocprofv3 -L > gpurun_out/$TAG/avail.txt 2>&1 || true
grep -o "SQ_INSTS_[A-Z0-9_]*\|SQ_[A-Z_]*MFMA[A-Z0-9_]*\|SQ_ACTIVE_INST_[A-Z_]*\|SQ_INST_CYCLES_[A-Z_]*\|SQ_BUSY_CYCLES\|SQ_WAVE_CYCLES\|SQ_WAIT_INST_ANY\|SQ_VALU_[A-Z_]*\|GRBM_GUI_ACTIVE" gpurun_out/$TAG/avail.txt | sort -u > gpurun_out/$TAG/avail_sq.txt || true
echo listed
pass() {   # pass <name> <counters...>
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/$TAG/$name -- python bench.py --steps 2 --warmup 10 --no-cpu-baseline --no-to-tol > gpurun_out/$TAG/$name.log 2>&1 || { echo "$name FAILED"; tail -3 gpurun_out/$TAG/$name.log; return 0; }
  local args=""
  for c in "$@"; do args="$args $c=gpurun_out/$TAG/$name"; done
  python tools/pmc_summary.py $args --prefix aq_core > gpurun_out/$TAG/instmix_$name.txt
  find gpurun_out/$TAG/$name -name "*kernel_trace.csv" -delete
  find gpurun_out/$TAG/$name -name "*counter_collection.csv" -delete
  echo $name done
}
want() { [ -z "$AQ_MIX_PASSES" ] || [[ " $AQ_MIX_PASSES " == *" $1 "* ]]; }   # AQ_MIX_PASSES="a b h": only those passes
want a && pass a SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS
want b && pass b SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES
want c && pass c SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64
want d && pass d SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT
want e && pass e SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY
want f && pass f SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS
want g && pass g SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
# memory side: average latency of a vector memory instruction = SQ_INST_LEVEL_VMEM / SQ_INSTS_VMEM (cycles), L2 hit rate, L1 -> L2 read latency
want h && pass h SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS
want i && pass i TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
want j && pass j TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
cat gpurun_out/$TAG/instmix_*.txt
