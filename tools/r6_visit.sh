#!/bin/bash
# scratch driver of a GPU-box visit (round 6); edited per visit
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=gpurun_out/r6an; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_hip_history.py -q -x -k "beyond_2_gib" 2>&1 | tail -8 | cut -c1-400
