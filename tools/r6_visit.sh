#!/bin/bash
# scratch driver of a GPU-box visit (round 6); edited per visit
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=gpurun_out/r6bf; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_hip_model.py tests/test_hip_eval_geometry.py tests/test_hip_00_rccl.py -q -x 2>&1 | tail -4 | cut -c1-300
S=$SECONDS
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $? in $((SECONDS-S)) s"
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r6bf/bench.json").read().strip().splitlines()[-1])
c = r["config"]
print(r["value"], r["ms_per_step"], {k: c[k] for k in c if k.startswith(("c3_", "c4_"))})
PY
