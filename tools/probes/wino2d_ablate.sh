#!/bin/bash
# Ablation builds of tools/probes/winograd2d.hip (RPG_W2_ABL bit mask: 1 no input loads | 2 no weight loads | 4 no stage writes |
# 8 no barrier | 16 no transform arithmetic; wrong results, timing only).  Run in the BUILD container from the repo root:
#     tools/probes/wino2d_ablate.sh 1 2 3 4 8 16 31      -> relpose-gnn_amd/lib/abl_w2_<mask>.so
# then on the GPU box:  RPG_LIB_PATH=relpose-gnn_amd/lib/abl_w2_3.so python tools/conv_bench.py --only l3.c1 --wino2d-ab ...
set -e
L=relpose-gnn_amd/lib
for m in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DRPG_PROBE_WINO2D -DRPG_W2_ABL=$m -c relpose-gnn_amd/csrc/winograd.hip -o /tmp/w2_abl_$m.o
  objs=$(ls $L/*.o | grep -v "/winograd.o")
  hipcc --offload-arch=gfx950 -shared -fPIC -o $L/abl_w2_$m.so $objs /tmp/w2_abl_$m.o
  echo built $L/abl_w2_$m.so
done
