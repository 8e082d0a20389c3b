#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=gpurun_out/r6h; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_bf16.py -q -m gpu -x  > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log; tail -6 $OUT/pytest.log
#timeout 300 python tools/conv_bench.py --bf16 --block64 --nimg 512 --warm 3 > $OUT/block64_56.txt 2>&1; grep block64 $OUT/block64_56.txt
#timeout 300 python tools/conv_bench.py --bf16 --block64 --nimg 512 --warm 3 --block64-map 64x86 > $OUT/block64_86.txt 2>&1; grep block64 $OUT/block64_86.txt
