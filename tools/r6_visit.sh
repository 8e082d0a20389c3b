#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=gpurun_out/r6t; mkdir -p $OUT
for t in 1 4 1 4; do
timeout 300 python tools/conv_bench.py --bf16 --nimg 512 --warm 3 --only l3.c --tune 17=$t 2>&1 | grep conv
timeout 300 python tools/conv_bench.py --bf16 --nimg 512 --warm 3 --only l4.c --tune 17=$t 2>&1 | grep conv
done
