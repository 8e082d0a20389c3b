#!/bin/bash
OUT=$1
cd "${GRAFT_REPO_ROOT:-.}"
timeout 600 python -m pytest tests/test_hip_bf16.py tests/test_hip_bench_geometry.py tests/test_hip_eval_geometry.py tests/test_hip_model.py -q -m gpu -x -k "bf16" > "$OUT/pytest_bf16.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest_bf16.log"
for i in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 5 --cpu-baseline-seconds 0 --no-other-configs --graphs 64 --encoder-dtype bf16 --gnn-dtype bf16 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read()); print(l['value'], l['ms_per_step'], l['roofline']['frac'], l['roofline']['avg_launch_ms'])"
done | tee "$OUT/bench_bf16.txt"
