#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=gpurun_out/r6g; mkdir -p $OUT
timeout 600 python tools/fixup_prio_ab.py 2 > $OUT/fixup_prio_ab.txt 2>&1; grep -v amdgpu.ids $OUT/fixup_prio_ab.txt
timeout 600 python tools/fixup_prio_ab.py 1 >> $OUT/fixup_prio_ab.txt 2>&1; grep -v amdgpu.ids $OUT/fixup_prio_ab.txt | tail -3
export TMPDIR=/tmp; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6g/prof_prio -o t -- python3 $R/bench.py --steps 10 --warmup 3 --cpu-baseline-seconds 0 --no-other-configs --no-kernel-timing --no-latency --tune 30=1 > $R/$OUT/prof_prio.log 2>&1
cd $R; python tools/rocprof_summary.py $(ls gpurun_out/r6g/prof_prio/*/*kernel_stats.csv | head -1) > $OUT/kernel_stats_prio_2streams.txt 2>&1; head -14 $OUT/kernel_stats_prio_2streams.txt
rm -rf gpurun_out/r6g/prof_prio
