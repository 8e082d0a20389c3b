#!/bin/bash
OUT=$1
cd "${GRAFT_REPO_ROOT:-.}"
timeout 1200 python -m pytest tests/test_hip_bf16.py tests/test_hip_bench_geometry.py tests/test_hip_eval_geometry.py -q -m gpu -k "bf16" > "$OUT/pytest_bf16.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest_bf16.log"
timeout 300 python tools/conv_bench.py --bf16 --nimg 512 --warm 3 --reps 10 --only "l" 2>&1 | grep "^conv" | tee "$OUT/conv_bf16_512.txt"
for i in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 5 --cpu-baseline-seconds 0 --no-other-configs --graphs 64 --encoder-dtype bf16 --gnn-dtype bf16 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read()); print(l['value'], l['ms_per_step'], l['roofline']['frac'], l['roofline']['avg_launch_ms'])"
done | tee "$OUT/bench_bf16.txt"
