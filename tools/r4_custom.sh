#!/bin/bash
OUT=$1
cd "${GRAFT_REPO_ROOT:-.}"
for st in 0 100 200 400 800 0; do
  echo "== stagger $st ticks (10 ns)"
  for mode in 1 2; do
    timeout 300 python tools/conv_bench.py --bf16 --nimg 512 --warm 3 --reps 10 --only "l1.c" --tune 17=$mode --tune 23=$st 2>&1 | grep "^conv" | sed "s/^/patch$mode /"
  done
  timeout 300 python tools/conv_bench.py --bf16 --nimg 512 --warm 3 --reps 10 --only "l2.c" --tune 23=$st 2>&1 | grep "^conv"
  timeout 300 python tools/conv_bench.py --bf16 --nimg 512 --warm 3 --reps 10 --only "l3.c1" --tune 23=$st 2>&1 | grep "^conv"
done | tee "$OUT/stagger.txt"
