#!/bin/bash
# scratch step of tools/r4_visit.sh (the last experiment run through it: stream-count sweep of the bf16 configuration --
# 1 / 2 / 3 / 4 streams at 64 graphs: 12.16 / 13.23 / 12.62 / 9.97 k graphs/s; at 128 graphs 2 / 3 / 4: 13.83 / 13.61 / 12.40)
OUT=$1
echo "custom: nothing to run"
