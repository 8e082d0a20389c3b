#!/bin/bash
# One GPU-box visit: the full -m gpu suite (or the ids given), then bench.py; logs under gpurun_out/<tag>/.
# usage: tools/gpu_check.sh <tag> [pytest args...]
TAG=${1:-check}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
rm -f gpurun_out/parity_report.jsonl
if [ $# -gt 0 ]; then
  timeout 1500 python -m pytest "$@" -q -m gpu -x > "$OUT/pytest.log" 2>&1
else
  timeout 1800 python -m pytest tests -q -m gpu -x > "$OUT/pytest.log" 2>&1
fi
echo "pytest rc=$?" >> "$OUT/pytest.log"
tail -5 "$OUT/pytest.log"
cp gpurun_out/parity_report.jsonl "$OUT/" 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench rc=$?"
cut -c1-1500 "$OUT/bench.json"
