#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=gpurun_out/r6s; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_hip_ops.py tests/test_hip_bench_geometry.py -q -m gpu -x > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
