#!/bin/bash
# build variants of stem.hip with -D flags into separate .so files and time them
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/relpose-gnn_amd/lib
for v in "" "-DST_NO_EPI" "-DST_NO_PATCH"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $v -c ../csrc/stem.hip -o /tmp/stem_v.o 2>/dev/null || { echo build failed $v; continue; }
  hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libv.so /tmp/stem_v.o conv_bf16.o encoder_ops.o forward.o gemm_f32.o gnn_ops.o timing.o winograd.o
  echo "== variant [$v]"; RPG_HIP_LIB=/tmp/libv.so python $R/tools/stem_bench.py
done
