#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=gpurun_out/r6i; mkdir -p $OUT
bash tools/gpu_check.sh r6i > $OUT/gpu_check.log 2>&1; tail -8 $OUT/gpu_check.log | cut -c1-600
bash tools/collect_profiles.sh r6 > $OUT/collect.log 2>&1; tail -5 $OUT/collect.log
