#!/usr/bin/env python3
"""Per-launch-geometry kernel times from a rocprofv3 --kernel-trace CSV: average duration of every (kernel, grid size)
pair, i.e. one line per ResNet stage for the convolution kernels.  usage: per_layer_times.py <dir with *_kernel_trace.csv> [substr]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
acc = defaultdict(list)
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            name = r["Kernel_Name"]
            if sub and sub not in name:
                continue
            g = int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1))))
            acc[(name.split("(")[0][-60:], g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
tot = sum(sum(v) for v in acc.values())
for (n, g), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print(f"{n:60s} wgs {g:6d}  n {len(v):5d}  avg {sum(v)/len(v):9.1f} us  med {v2[len(v2)//2]:9.1f}  min {v2[0]:9.1f}  share {100*sum(v)/tot:5.1f} %")
