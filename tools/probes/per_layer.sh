#!/bin/bash
# Per-stage (grid size) times and MFMA occupancy of the Winograd kernel inside the model (GPU box).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/per_layer_${1:-x}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
B="$R/bench.py --steps 5 --warmup 2 --cpu-baseline-seconds 0 --no-kernel-timing"
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/t1 -o t -- python3 $B --streams 1 > $OUT/t1.log 2>&1
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/t2 -o t -- python3 $B > $OUT/t2.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_F32 --output-format csv -d $OUT/p1 -o p -- python3 $B --streams 1 > $OUT/p1.log 2>&1
cd $R
echo "== 1 stream" > $OUT/summary.txt
python3 tools/probes/per_layer_times.py $OUT/t1 >> $OUT/summary.txt 2>&1
echo "== 2 streams" >> $OUT/summary.txt
python3 tools/probes/per_layer_times.py $OUT/t2 >> $OUT/summary.txt 2>&1
echo "== pmc 1 stream" >> $OUT/summary.txt
python3 tools/probes/per_layer_pmc.py $OUT/p1 wino43_conv8 >> $OUT/summary.txt 2>&1
rm -rf $OUT/t1 $OUT/t2 $OUT/p1
cat $OUT/summary.txt
