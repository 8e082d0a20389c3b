#!/bin/bash
OUT=$1
cd "${GRAFT_REPO_ROOT:-.}"
# exact-round geometry for the nested kernel (no tile-quantisation loss): layer 3 at 512 images = 1792 workgroups = 7 rounds,
# layer 2 at 512 images = 3136 = 12.25 rounds, layer 1 at 256 = 3136
for n in 512 256; do
  echo "== images $n"
  timeout 600 python tools/conv_bench.py --nimg $n --warm 3 --reps 10 --only "c1" --wino2d-ab 2>&1 | grep "^conv" | cut -c1-175
done | tee "$OUT/wino2d_ab_exact_rounds.txt"
