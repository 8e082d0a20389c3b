#!/bin/bash
# scratch driver of a GPU-box visit (round 6); edited per visit
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=gpurun_out/r6at; mkdir -p $OUT
S=$SECONDS
bash tools/collect_profiles.sh r6final > $OUT/collect.log 2>&1; echo "collect rc $? in $((SECONDS-S)) s"; tail -14 $OUT/collect.log | cut -c1-200
S=$SECONDS
timeout 400 python tools/soak.py --seconds 240 > $OUT/soak.json 2> $OUT/soak.err; echo "soak rc $? in $((SECONDS-S)) s"; cat $OUT/soak.json
S=$SECONDS
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $? in $((SECONDS-S)) s"
tail -1 $OUT/bench.json | cut -c1-1500
