#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=gpurun_out/r6w; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_bf16.py tests/test_hip_random_shapes.py -q -m gpu -x -k "stem" > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
timeout 300 python tools/stem_bench.py 512 --bf16 > $OUT/stem_bench.txt 2>&1; grep -v amdgpu.ids $OUT/stem_bench.txt
