#!/bin/bash
# Round-end profile collection on the GPU box (run from the repo root):  tools/collect_profiles.sh <tag>
# 1. rocprofv3 --kernel-trace --stats of the default bench command and of the single-stream variant
# 2. PMC passes (counters only) for HBM traffic of the single-stream variant: FETCH_SIZE and WRITE_SIZE separately
set -u
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
B="$R/bench.py --steps 5 --warmup 2 --cpu-baseline-seconds 0"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_default" -o t -- python3 $B > "$OUT/trace_default.log" 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_1stream" -o t -- python3 $B --streams 1 > "$OUT/trace_1stream.log" 2>&1
P="$R/bench.py --steps 3 --warmup 1 --cpu-baseline-seconds 0 --streams 1 --no-kernel-timing"
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o p -- python3 $P > "$OUT/pmc_fetch.log" 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o p -- python3 $P > "$OUT/pmc_write.log" 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_F32 --output-format csv -d "$OUT/pmc_mfma" -o p -- python3 $P > "$OUT/pmc_mfma.log" 2>&1
# keep the merge small: drop the per-dispatch counter rows once summarised is not possible here; just list
du -sh "$OUT"/* | tail -12
grep -h '"metric"' "$OUT"/trace_default.log "$OUT"/trace_1stream.log | cut -c1-200
