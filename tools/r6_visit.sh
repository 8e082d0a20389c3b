#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=gpurun_out/r6y; mkdir -p $OUT
for n in 512 256 128 64; do timeout 300 python tools/stem_bench.py $n --bf16 --variants "strips, default" 2>&1 | grep "fp32 in"; done
timeout 300 python tools/stem_bench.py 512 --bf16 --shape 256x341 --variants "strips, default" 2>&1 | grep "fp32 in"
