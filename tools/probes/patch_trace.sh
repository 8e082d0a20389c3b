#!/bin/bash
# Phase trace of the bf16 patch kernel (csrc/conv_bf16.hip, -DRPG_PATCH_TRACE: eight s_memtime stamps per workgroup, medians printed per launch).
# Build container, repo root:  tools/probes/patch_trace.sh   -> relpose-gnn_amd/lib/trace_patch.so
# GPU box:  RPG_PATCH_TRACE=1 RPG_LIB_PATH=$PWD/relpose-gnn_amd/lib/trace_patch.so python tools/conv_bench.py --bf16 --nimg 512 --reps 3 --only l
set -e
L=relpose-gnn_amd/lib
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DRPG_PATCH_TRACE -c relpose-gnn_amd/csrc/conv_bf16.hip -o /tmp/patch_trace.o
objs=$(ls $L/*.o | grep -v "/conv_bf16.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $L/trace_patch.so $objs /tmp/patch_trace.o
echo built $L/trace_patch.so
