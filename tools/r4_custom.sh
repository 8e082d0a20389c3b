#!/bin/bash
# scratch step of tools/r4_visit.sh
OUT=$1
timeout 600 python -m pytest tests/test_hip_bf16.py -q -m gpu -x -k "patch_kernel" > "$OUT/pytest_patch.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest_patch.log"
for mode in 2 4; do
  echo "== RPG_TUNE_BF16_PATCH=$mode"
  timeout 300 python tools/conv_bench.py --bf16 --nimg 512 --warm 3 --reps 10 --only l --tune 17=$mode 2>&1 | grep -E "l1|l2" | head -12
done
for mode in 1 4; do
  timeout 300 python bench.py --steps 20 --warmup 5 --cpu-baseline-seconds 0 --no-other-configs --no-latency --graphs 64 --encoder-dtype bf16 --gnn-dtype bf16 --tune 17=$mode > "$OUT/b_$mode.json" 2> "$OUT/b_$mode.err"
  python3 - "$OUT/b_$mode.json" $mode <<'PY'
import json,sys
l=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], l['value'], l['ms_per_step'], l['roofline']['frac'])
PY
done
