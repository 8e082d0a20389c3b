#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=gpurun_out/r6b; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_hip_bf16.py tests/test_hip_model.py tests/test_hip_eval_geometry.py tests/test_hip_bench_geometry.py -q -m gpu -x > $OUT/pytest_bf16.log 2>&1; echo "rc=$?" >> $OUT/pytest_bf16.log; tail -15 $OUT/pytest_bf16.log
#timeout 300 python tools/stem_bench.py 512 --bf16 > $OUT/stem_bench.txt 2>&1; grep -v amdgpu.ids $OUT/stem_bench.txt | head -6
#timeout 300 python tools/stem_bench.py 512 --bf16 --shape 256x341 > $OUT/stem_bench_341.txt 2>&1; grep -v amdgpu.ids $OUT/stem_bench_341.txt | head -6
#timeout 300 python tools/stem_bench.py 64 --bf16 > $OUT/stem_bench_64.txt 2>&1; grep -v amdgpu.ids $OUT/stem_bench_64.txt | head -4
#timeout 300 python tools/stem_bench.py 8 --bf16 > $OUT/stem_bench_8.txt 2>&1; grep -v amdgpu.ids $OUT/stem_bench_8.txt | head -4
