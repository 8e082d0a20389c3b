#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=gpurun_out/r6aa; mkdir -p $OUT
timeout 300 python tools/conv_bench.py --bf16 --block64 --nimg 512 --warm 10 2>&1 | grep block64
RPG_LIB_PATH=$R/relpose-gnn_amd/lib/abl_block_1.so timeout 300 python tools/conv_bench.py --bf16 --block64 --nimg 512 --warm 10 2>&1 | grep block64
