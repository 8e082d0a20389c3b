#!/bin/bash
# scratch step of tools/r4_visit.sh: phase trace of the bf16 patch kernel (tools/probes/patch_trace.sh)
OUT=$1
RPG_PATCH_TRACE=1 RPG_LIB_PATH=$PWD/relpose-gnn_amd/lib/trace_patch.so timeout 600 python tools/conv_bench.py --bf16 --nimg 512 --warm 2 --reps 3 --only .c > "$OUT/patch_trace.txt" 2>&1
grep "patch_trace" "$OUT/patch_trace.txt" | awk '(++n % 5 == 0)' | cut -c1-420
