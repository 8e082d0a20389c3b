#!/bin/bash
# As wino_ablate.sh, but runs tools/bench_brief.py (the whole model) against the ablation build:
#   tools/probes/wino_ablate_bench.sh "<-D flags>" [bench args...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
FLAGS=$1; shift
cd $R/relpose-gnn_amd/lib
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -c ../csrc/winograd.hip -o /tmp/wino_abl.o 2>/dev/null || { echo build failed; exit 1; }
hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libabl.so /tmp/wino_abl.o conv_bf16.o encoder_ops.o forward.o gemm_f32.o gnn_ops.o timing.o stem.o
cd $R
echo "== flags: $FLAGS"
RPG_HIP_LIB=/tmp/libabl.so python tools/bench_brief.py "$@" 2>&1 | grep "^value"
