#!/bin/bash
# Profile collection on the GPU box (run from the repo root):  tools/collect_profiles.sh <tag>
# 1. rocprofv3 --kernel-trace --stats of the default bench command and of the single-stream variant
# 2. PMC passes (counters only, no other tracing) of the single-stream variant: FETCH_SIZE, WRITE_SIZE, MFMA occupancy
set -u
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
B="$R/bench.py --steps 5 --warmup 2 --cpu-baseline-seconds 0 --no-latency --no-other-configs"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_default" -o t -- python3 $B > "$OUT/trace_default.log" 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_1stream" -o t -- python3 $B --streams 1 > "$OUT/trace_1stream.log" 2>&1
P="$R/bench.py --steps 3 --warmup 1 --cpu-baseline-seconds 0 --no-other-configs --streams 1 --no-kernel-timing"
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o p -- python3 $P > "$OUT/pmc_fetch.log" 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o p -- python3 $P > "$OUT/pmc_write.log" 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_F32 --output-format csv -d "$OUT/pmc_mfma" -o p -- python3 $P > "$OUT/pmc_mfma.log" 2>&1
# bf16 path (configs[2]: 64 graphs, bf16 encoder + bf16 GNN Linears): kernel trace + matrix-pipe / LDS / traffic counters
BB="$R/bench.py --steps 5 --warmup 2 --cpu-baseline-seconds 0 --graphs 64 --encoder-dtype bf16 --gnn-dtype bf16"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_bf16_1stream" -o t -- python3 $BB --streams 1 > "$OUT/trace_bf16_1stream.log" 2>&1
PB="$BB --steps 3 --warmup 1 --streams 1 --no-kernel-timing"
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d "$OUT/pmc_bf16_sq" -o p -- python3 $PB > "$OUT/pmc_bf16_sq.log" 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_BF16 --output-format csv -d "$OUT/pmc_bf16_mops" -o p -- python3 $PB > "$OUT/pmc_bf16_mops.log" 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_bf16_fetch" -o p -- python3 $PB > "$OUT/pmc_bf16_fetch.log" 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_bf16_write" -o p -- python3 $PB > "$OUT/pmc_bf16_write.log" 2>&1
cd "$R"
for k in block64_bf16_fused conv3x3_bf16_patch conv_bf16_dma stem_strip_bf16; do
  python3 tools/pmc_summary.py "$OUT" $k >> "$OUT/pmc_bf16_summary.txt" 2>&1
done
f=$(ls "$OUT"/trace_bf16_1stream/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && python3 tools/rocprof_summary.py "$f" "$OUT/kernel_stats_bf16_1stream.txt" > /dev/null
rm -rf "$OUT"/pmc_bf16_sq "$OUT"/pmc_bf16_mops "$OUT"/pmc_bf16_fetch "$OUT"/pmc_bf16_write "$OUT"/trace_bf16_1stream/*kernel_trace.csv
python3 tools/pmc_traffic.py "$OUT" "wino43_conv8" "$OUT/pmc_wino43.json" 32 > "$OUT/pmc_wino43.txt" 2>&1
python3 tools/pmc_summary.py "$OUT" wino43 > "$OUT/pmc_wino43_summary.txt" 2>&1
python3 tools/pmc_summary.py "$OUT" stem_pool > "$OUT/pmc_stem_summary.txt" 2>&1
for t in trace_default trace_1stream; do
  f=$(ls "$OUT/$t"/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 tools/rocprof_summary.py "$f" "$OUT/kernel_stats_$t.txt" > /dev/null
done
# the per-dispatch counter rows are large: keep the summaries only
rm -rf "$OUT"/pmc_fetch "$OUT"/pmc_write "$OUT"/pmc_mfma "$OUT"/trace_default/*kernel_trace.csv "$OUT"/trace_1stream/*kernel_trace.csv
du -sh "$OUT" | tail -1
cat "$OUT/pmc_wino43.txt" | tail -30
head -12 "$OUT/kernel_stats_trace_1stream.txt"
