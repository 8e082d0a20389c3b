#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=gpurun_out/r6m; mkdir -p $OUT
for rep in 1 2 3; do for t in 1 5; do
timeout 300 python bench.py --graphs 64 --encoder-dtype bf16 --gnn-dtype bf16 --steps 20 --warmup 5 --cpu-baseline-seconds 0 --no-other-configs --no-latency --tune 28=$t > $OUT/b.json 2>/dev/null
python - <<PY
import json
d=json.load(open('$OUT/b.json'))
print('tail=$t', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline']['launches'])
PY
done; done
timeout 300 python tools/conv_bench.py --bf16 --nimg 512 --warm 3 --only l2.c --tune 28=5 2>&1 | grep conv
timeout 300 python tools/conv_bench.py --bf16 --nimg 512 --warm 3 --only l2.c --tune 28=1 2>&1 | grep conv
