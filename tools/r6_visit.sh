#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=gpurun_out/r6n; mkdir -p $OUT
timeout 600 python tools/soak.py --seconds 240 > $OUT/soak.json 2> $OUT/soak.err; tail -c 1500 $OUT/soak.json
