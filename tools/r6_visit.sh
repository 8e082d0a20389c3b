#!/bin/bash
# scratch driver of a GPU-box visit (round 6); edited per visit
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=gpurun_out/r6bc; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mlpb tools/probes/mfma_lds_power_probe_bf16.hip 2>/dev/null
timeout 600 /tmp/mlpb 2>&1 | tee $OUT/mfma_lds_power_bf16.txt
