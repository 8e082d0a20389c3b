#!/bin/bash
# Ablation builds of the bf16 patch kernel (csrc/conv_bf16.hip, RPG_PATCH_ABL: 1 no epilogue | 2 no per-step wait + barrier | 4 epilogue stores confined to a 64-KB window; wrong
# results, timing only).  Build container, repo root:  tools/probes/patch_ablate.sh 1 2 3  -> relpose-gnn_amd/lib/abl_patch_<m>.so
# GPU box:  RPG_LIB_PATH=$PWD/relpose-gnn_amd/lib/abl_patch_1.so python tools/conv_bench.py --bf16 --nimg 512 --only l1.c --tune 17=2
set -e
L=relpose-gnn_amd/lib
for m in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DRPG_PATCH_ABL=$m -c relpose-gnn_amd/csrc/conv_bf16.hip -o /tmp/patch_abl_$m.o
  objs=$(ls $L/*.o | grep -v "/conv_bf16.o")
  hipcc --offload-arch=gfx950 -shared -fPIC -o $L/abl_patch_$m.so $objs /tmp/patch_abl_$m.o
  echo built $L/abl_patch_$m.so
done
