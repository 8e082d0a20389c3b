#!/bin/bash
# Collect PMC counters for one workload in separate rocprofv3 passes (counters only: no sys/hip/hsa tracing).
# usage: tools/pmc_run.sh <outdir-under-gpurun_out> <python args...>     (run from the repo root on the GPU box)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
pass() {  # name, counters...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- python3 "$R/$PYSCRIPT" $PYARGS > "$OUT/$name.log" 2>&1
}
PYSCRIPT=$1; shift
PYARGS="$*"
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 GRBM_GUI_ACTIVE
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_WAVES
pass tcc1 FETCH_SIZE
pass tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
pass tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr
ls -R "$OUT" | head -40
