#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files under a directory: per kernel name, mean counter value per
dispatch (and duration from the kernel trace).  usage: tools/pmc_summary.py gpurun_out/pmc_l2 [substr]"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "gemm_tile"
for f in sorted(glob.glob(os.path.join(root, "*", "*counter_collection.csv"))):
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        if sub not in r["Kernel_Name"]:
            continue
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = defaultdict(list)
    kt = f.replace("counter_collection", "kernel_trace")
    if os.path.exists(kt):
        for r in csv.DictReader(open(kt)):
            if sub in r["Kernel_Name"]:
                dur[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("==", os.path.basename(os.path.dirname(f)))
    for k, cs in acc.items():
        d = dur.get(k, [0])
        print(f"  {k}  dispatches={len(next(iter(cs.values())))}  avg_us={sum(d)/len(d):.1f}")
        for c, v in cs.items():
            print(f"     {c:32s} {sum(v)/len(v):16.1f}")
