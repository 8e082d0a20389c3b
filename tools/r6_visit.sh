#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=gpurun_out/r6v; mkdir -p $OUT
timeout 300 python tools/conv_bench.py --warm 3 --only gtp > $OUT/gtp2.txt 2>&1; grep lin $OUT/gtp2.txt
timeout 300 python tools/conv_bench.py --warm 3 --only gtp >> $OUT/gtp2.txt 2>&1; grep lin $OUT/gtp2.txt | tail -8
