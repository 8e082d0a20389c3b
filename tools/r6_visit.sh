#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=gpurun_out/r6z; mkdir -p $OUT
bash tools/gpu_check.sh r6z > $OUT/gpu_check.log 2>&1; tail -6 $OUT/gpu_check.log | cut -c1-300
bash tools/collect_profiles.sh r6c > $OUT/collect.log 2>&1; tail -3 $OUT/collect.log
