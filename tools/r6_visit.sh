#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=gpurun_out/r6k; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_bf16.py -q -m gpu -x -k "paired or block_fusion" > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log; tail -5 $OUT/pytest.log
for rep in 1 2 3; do for t in 1 0; do
timeout 300 python bench.py --graphs 64 --encoder-dtype bf16 --gnn-dtype bf16 --steps 20 --warmup 5 --cpu-baseline-seconds 0 --no-other-configs --no-latency --tune 31=$t > $OUT/b_$t_$rep.json 2>/dev/null
python - <<PY
import json
d=json.load(open('$OUT/b_$t_$rep.json'))
print('pair=$t', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline']['launches'])
PY
done; done
