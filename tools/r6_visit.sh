#!/bin/bash
# scratch driver of a GPU-box visit (round 6); edited per visit
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=$R/gpurun_out/r6ar; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lat1 -o t -- python3 $R/tools/probes/lat1_profile.py > $OUT/lat1.log 2>&1
cd $R
grep -v "amdgpu.ids\|simple_timer" $OUT/lat1.log | tail -5
f=$(find $OUT/lat1 -name "*kernel_stats.csv" | head -1); echo $f
python tools/rocprof_summary.py "$f" $OUT/lat1_kernel_stats.txt; head -45 $OUT/lat1_kernel_stats.txt | cut -c1-190
rm -rf $OUT/lat1
