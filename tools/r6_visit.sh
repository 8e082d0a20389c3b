#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=gpurun_out/r6f; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_bf16.py tests/test_hip_model.py -q -m gpu -x -k "fused_stem or lookahead or reference_eval_loop or pipeline or host_rounded" > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
timeout 600 python tools/lookahead_probe.py > $OUT/lookahead_probe.txt 2>&1; grep -v amdgpu.ids $OUT/lookahead_probe.txt
timeout 300 python tools/stem_bench.py 512 --bf16 --variants default > $OUT/stem_bench.txt 2>&1; grep -v amdgpu.ids $OUT/stem_bench.txt
