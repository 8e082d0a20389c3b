#!/bin/bash
# scratch driver of a GPU-box visit (round 6); edited per visit
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=gpurun_out/r6ag; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_bf16.py -x -q -k "basicblock64 or block_fusion" 2>&1 | tail -5
for t in "" "--tune 27=3" "" "--tune 27=3"; do
timeout 600 python bench.py --graphs 64 --encoder-dtype bf16 --gnn-dtype bf16 --steps 10 --warmup 3 --no-other-configs --no-latency --cpu-baseline-seconds 0 $t 2>/dev/null | tail -1 | python -c "
import sys, json; r = json.loads(sys.stdin.read()); print('$t', r['value'], r['ms_per_step'], r['roofline'].get('frac'))"
done
