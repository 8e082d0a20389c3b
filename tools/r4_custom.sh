#!/bin/bash
# scratch step of tools/r4_visit.sh (edited per visit)
OUT=$1
cd "${GRAFT_REPO_ROOT:-.}"
timeout 1200 python -m pytest tests/test_hip_ops.py -q -m gpu -x -k "last_arriver" > "$OUT/pytest_fix.log" 2>&1; echo "pytest rc=$?"; tail -5 "$OUT/pytest_fix.log"
for mode in 0 1 2 3 0 1; do
  echo "== in-kernel fixup mode $mode: configs[1]"; timeout 300 python bench.py --steps 20 --warmup 5 --cpu-baseline-seconds 0 --no-other-configs --tune 20=$mode 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read()); print(l['value'], l['ms_per_step'], l['roofline']['frac'], l['latency_1graph'])"
done
for mode in 0 1 2 3; do
  echo "== mode $mode: 1 graph"; timeout 300 python bench.py --steps 40 --warmup 5 --cpu-baseline-seconds 0 --no-other-configs --graphs 1 --streams 1 --no-kernel-timing --tune 20=$mode 2>/dev/null | cut -c1-200
done
