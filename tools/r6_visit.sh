#!/bin/bash
# scratch driver of a GPU-box visit (round 6); edited per visit
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=gpurun_out/r6bd; mkdir -p $OUT
timeout 500 python tools/soak.py --seconds 300 > $OUT/soak.json 2> $OUT/soak.err; echo "soak rc $?"; cat $OUT/soak.json; tail -3 $OUT/soak.err | grep -v amdgpu
