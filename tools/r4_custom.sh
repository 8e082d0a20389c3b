#!/bin/bash
# scratch step of tools/r4_visit.sh
OUT=$1
timeout 900 python -m pytest tests/test_hip_bf16.py -q -m gpu -x -k "linear_bf16" > "$OUT/pytest_lin.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest_lin.log"
for c in 0 10 11 12 13 17 18; do
  timeout 300 python bench.py --steps 20 --warmup 5 --cpu-baseline-seconds 0 --no-other-configs --no-latency --graphs 64 --encoder-dtype bf16 --gnn-dtype bf16 --tune 24=$c > "$OUT/lin_$c.json" 2> "$OUT/lin_$c.err"
  python3 - "$OUT/lin_$c.json" $c <<'PY'
import json,sys
l=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], l['value'], l['ms_per_step'], l['roofline']['frac'], json.dumps(l.get('other_kernels',{}).get('linear')))
PY
done
