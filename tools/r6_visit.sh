#!/bin/bash
# scratch driver of a GPU-box visit (round 6); edited per visit
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=gpurun_out/r6ba; mkdir -p $OUT
S=$SECONDS
timeout 3000 python -m pytest tests -q -x -m gpu 2>&1 | tail -4 | cut -c1-300
echo "gpu suite: $((SECONDS-S)) s"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
S=$SECONDS
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $? in $((SECONDS-S)) s"
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r6ba/bench.json").read().strip().splitlines()[-1])
c = r["config"]
print(r["value"], r["ms_per_step"], r["roofline"]["frac"], {k: c[k] for k in c if k.startswith(("c2_", "c3_", "c4_", "lat1"))})
PY
