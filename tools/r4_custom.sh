#!/bin/bash
OUT=$1
cd "${GRAFT_REPO_ROOT:-.}"
for mode in 1 2 1 2; do
  echo "== RPG_TUNE_BF16_PATCH=$mode: bench --graphs 64 bf16"
  timeout 300 python bench.py --steps 20 --warmup 5 --cpu-baseline-seconds 0 --no-other-configs --graphs 64 --encoder-dtype bf16 --gnn-dtype bf16 --tune 17=$mode 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read()); print(l['value'], l['ms_per_step'], l['roofline']['frac'], l['roofline']['avg_launch_ms'])"
done | tee "$OUT/patch_mode_bench.txt"
