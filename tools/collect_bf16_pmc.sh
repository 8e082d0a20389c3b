#!/bin/bash
# PMC passes of the bf16 one-stream step only (counters only, separate passes): tools/collect_bf16_pmc.sh <tag>
set -u
TAG=${1:-r5b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
PB="$R/bench.py --cpu-baseline-seconds 0 --graphs 64 --encoder-dtype bf16 --gnn-dtype bf16 --steps 3 --warmup 1 --streams 1 --no-kernel-timing"
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d "$OUT/pmc_bf16_sq" -o p -- python3 $PB > "$OUT/pmc_bf16_sq.log" 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_bf16_fetch" -o p -- python3 $PB > "$OUT/pmc_bf16_fetch.log" 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_bf16_write" -o p -- python3 $PB > "$OUT/pmc_bf16_write.log" 2>&1
cd "$R"
for k in block64_bf16_fused conv3x3_bf16_patch stem_pool_bf16; do
  python3 tools/pmc_summary.py "$OUT" $k >> "$OUT/pmc_bf16_summary.txt" 2>&1
done
rm -rf "$OUT"/pmc_bf16_sq "$OUT"/pmc_bf16_fetch "$OUT"/pmc_bf16_write
cat "$OUT/pmc_bf16_summary.txt" | cut -c1-160 | head -80
