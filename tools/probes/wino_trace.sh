#!/bin/bash
# Build the library with the Winograd timeline instrumentation and run tools/probes/wino_trace.py (GPU box).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/relpose-gnn_amd/lib
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DRPG_WINO_TRACE -c ../csrc/winograd.hip -o /tmp/wino_tr.o 2>/dev/null || { echo build failed; exit 1; }
hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libtr.so /tmp/wino_tr.o conv_bf16.o encoder_ops.o forward.o gemm_f32.o gnn_ops.o timing.o stem.o stem_bf16.o
RPG_HIP_LIB=/tmp/libtr.so python $R/tools/probes/${SCRIPT:-wino_trace.py}
