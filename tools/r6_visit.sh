#!/bin/bash
# scratch driver of a GPU-box visit (round 6); edited per visit
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=gpurun_out/r6as; mkdir -p $OUT
timeout 1500 python tools/probes/big_launch.py 480x640x64 300x400x96 128x171x256 96x128x320 250x333x96 2>&1 | grep -v amdgpu.ids | tee $OUT/big_launch_shapes.txt | cut -c1-330
