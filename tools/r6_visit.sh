#!/bin/bash
# scratch driver of a GPU-box visit (round 6); edited per visit
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=gpurun_out/r6bb; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_bf16.py -q -x -k "matrix_pipe or block_fusion or large_batches" 2>&1 | tail -3
S=$SECONDS
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $? in $((SECONDS-S)) s"
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r6bb/bench.json").read().strip().splitlines()[-1])
c = r["config"]
print(r["value"], r["ms_per_step"], r["roofline"]["frac"], {k: c[k] for k in c if k.startswith(("c2_", "bf16_pipe"))})
print(json.dumps(r["other_configs"].get("bf16_matrix_pipe_sustained_pflops"))[:400])
PY
