#!/usr/bin/env python3
"""How the two streams of the default bench overlap: from a rocprofv3 --kernel-trace CSV, the share of the timed window with
0 / 1 / 2+ kernels in flight, and for the one-kernel periods which kernel was running alone (its launches do not fill the
chip by themselves if it is a small grid), plus the periods with fewer than 256 workgroups in flight.  Caveat: kernel tracing
slows the host's launches (12.6 instead of 10.5 ms per step in one run), so one stream can fall behind the other and kernels
run alone that overlap in the un-profiled run; two runs of this tool gave 77 % and 43 % "two or more".
usage: overlap_timeline.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import sys
from collections import defaultdict

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
ev = []
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# inside the timed region: launches 35 % .. 75 % (after warm-up / weight packing, before the post-bench passes)
rows = rows[int(len(rows) * 0.35):int(len(rows) * 0.75)]
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
for r in rows:
    import re
    m = re.search(r"([A-Za-z_][A-Za-z0-9_]*)(<[^(]*>)?\(", r["Kernel_Name"].replace("(anonymous namespace)::", ""))
    name = (m.group(1) if m else r["Kernel_Name"])[:40]
    wgs = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) if "Grid_Size_X" in r else int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"]))
    ev.append((int(r["Start_Timestamp"]), 1, name, wgs))
    ev.append((int(r["End_Timestamp"]), -1, name, wgs))
ev.sort()
active, last = {}, t0
dur = defaultdict(float)
alone = defaultdict(float)
small = defaultdict(float)            # periods in which the kernels in flight have < 256 workgroups between them
for t, d, name, wgs in ev:
    n = len(active)
    dur[min(n, 2)] += t - last
    if n == 1:
        alone[next(iter(active.values()))] += t - last
    if n >= 1 and sum(v[1] for v in active.values()) < 256:
        small[" + ".join(sorted(f"{v[0]}[{v[1]}]" for v in active.values()))] += t - last
    last = t
    key = (name, wgs)
    if d > 0:
        active[id(key) + t] = key
        ev_key = id(key) + t
    else:
        for k, v in list(active.items()):
            if v == key:
                del active[k]
                break
tot = t1 - t0
print(f"window {tot/1e6:.2f} ms: idle {100*dur[0]/tot:.1f} %, one kernel {100*dur[1]/tot:.1f} %, two or more {100*dur[2]/tot:.1f} %")
print("alone (share of the window, kernel, workgroups):")
for (name, wgs), v in sorted(alone.items(), key=lambda kv: -kv[1])[:10]:
    print(f"  {100*v/tot:5.1f} %  {name:40s} {wgs}")
print(f"fewer than 256 workgroups in flight: {100*sum(small.values())/tot:.1f} % of the window:")
for k, v in sorted(small.items(), key=lambda kv: -kv[1])[:12]:
    print(f"  {100*v/tot:5.2f} %  {k}")
