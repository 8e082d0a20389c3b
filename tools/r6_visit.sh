#!/bin/bash
# scratch driver of a GPU-box visit (round 6); edited per visit
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=gpurun_out/r6az; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mppb tools/probes/mfma_power_probe_bf16.hip 2>/dev/null
timeout 600 /tmp/mppb 2>&1 | tee $OUT/mfma_power_bf16.txt
