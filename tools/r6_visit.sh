#!/bin/bash
# scratch driver of a GPU-box visit (round 6); edited per visit
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=gpurun_out/r6aj; mkdir -p $OUT
echo "== current library"; timeout 1500 python -m pytest tests/test_hip_history.py -q -x 2>&1 | tail -8 | cut -c1-400
echo "== probe library with the old division (expected: failures on 256x341 bf16, 1 stream)"
RPG_HIP_LIB=$R/relpose-gnn_amd/lib/librelpose_gnn_hip_probe.so timeout 900 python -m pytest tests/test_hip_history.py -q -k "previous_forward and bf16" 2>&1 | tail -12 | cut -c1-300
