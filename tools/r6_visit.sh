#!/bin/bash
# scratch driver of a GPU-box visit (round 6); edited per visit
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=gpurun_out/r6ap; mkdir -p $OUT
timeout 600 python tools/probes/s2_bound.py 512 2>&1 | grep -v amdgpu.ids | tee $OUT/s2_bound.txt
