#!/bin/bash
# scratch driver of a GPU-box visit (round 6); edited per visit
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=gpurun_out/r6ax; mkdir -p $OUT
for i in 1 2; do
for v in "" 4 6; do
  lib=""; [ -n "$v" ] && lib=$R/relpose-gnn_amd/lib/librelpose_gnn_hip_probe$v.so
  RPG_HIP_LIB=$lib timeout 300 python tools/conv_bench.py --bf16 --block64 --nimg 512 --warm 10 2>&1 | grep "^block64" | tail -1 | sed "s#^#abl=${v:-0} #" | cut -c1-190
done; done | tee $OUT/block_abl.txt
