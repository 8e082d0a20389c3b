#!/bin/bash
# Scratch driver of a round-6 GPU-box visit (gpurun -- 'bash tools/r6_visit.sh'); edited per visit.  Last content: the final check.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=gpurun_out/r6_final; mkdir -p $OUT
timeout 3000 python -m pytest tests -q -x -m gpu 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 400 python tools/soak.py --seconds 240 > $OUT/soak.json 2> $OUT/soak.err; echo "soak rc $?"; cat $OUT/soak.json
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"; tail -1 $OUT/bench.json | cut -c1-300
