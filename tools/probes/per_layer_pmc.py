#!/usr/bin/env python3
"""Per-launch-geometry counters of one kernel from a rocprofv3 --pmc pass (counter_collection.csv + kernel_trace.csv in
the same directory): for every grid size (= ResNet stage) the mean duration and the mean of each counter, plus MFMA-busy
and CU-busy fractions when the SQ_* / GRBM counters are present.  usage: per_layer_pmc.py <pass dir> <kernel substr>"""
import csv
import glob
import os
import sys
from collections import defaultdict

root, sub = sys.argv[1], sys.argv[2]
f = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)[0]
grid = {}
cnt = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(f)):
    if sub not in r["Kernel_Name"]:
        continue
    g = int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"]))
    cnt[g][r["Counter_Name"]].append(float(r["Counter_Value"]))
    grid[r["Dispatch_Id"]] = g
dur = defaultdict(list)
kt = f.replace("counter_collection", "kernel_trace")
if os.path.exists(kt):
    for r in csv.DictReader(open(kt)):
        if r["Dispatch_Id"] in grid:
            dur[grid[r["Dispatch_Id"]]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for g in sorted(cnt):
    c = {k: sum(v) / len(v) for k, v in cnt[g].items()}
    d = dur.get(g, [0.0])
    line = f"wgs {g:6d}  n {len(next(iter(cnt[g].values()))):4d}  avg {sum(d)/len(d):8.1f} us"
    for k, v in c.items():
        line += f"  {k} {v:.4g}"
    if "GRBM_GUI_ACTIVE" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        simd_cycles = c["GRBM_GUI_ACTIVE"] * 1024          # per-XCD-summed? see pmc_traffic.py: 256 CUs x 4 SIMDs
        line += f"  | mfma_busy {c['SQ_VALU_MFMA_BUSY_CYCLES']/simd_cycles*8:.3f}"
        if "SQ_BUSY_CU_CYCLES" in c:
            line += f"  cu_busy {c['SQ_BUSY_CU_CYCLES']/(c['GRBM_GUI_ACTIVE']*256)*8:.3f}"
    print(line)
