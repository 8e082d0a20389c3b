#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=gpurun_out/r6j; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_random_shapes.py -q -m gpu -k "stem or basicblock" > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log; tail -15 $OUT/pytest.log
