#!/usr/bin/env python3
"""HBM traffic of the dominant kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE in separate runs).
Correction per MI355X_MICROARCH.md (HBM section): both counters are in KiB-like units of 1024 B
(hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024) and on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced
reads, so the read side is doubled.
usage: pmc_traffic.py <prof_dir> <kernel-substring> <out.json> [graphs_per_step [git_commit]]
The result is stamped with the digest of the kernel sources it was measured on (relpose_gnn_amd.build.source_digest) and
the batch size, which is how bench.py decides whether it may quote it."""
import csv
import json
import sys
from collections import defaultdict


def mean_counter(path, sub, name):
    vals = defaultdict(float)
    for r in csv.DictReader(open(path)):
        if sub in r["Kernel_Name"] and r["Counter_Name"] == name:
            vals[r["Dispatch_Id"]] += float(r["Counter_Value"])
    v = list(vals.values())
    return sum(v) / len(v), len(v)


def wino_algorithmic_bytes(n_img):
    """Compulsory HBM bytes of an average Winograd launch of the ResNet34 encoder at 224x224: input + output (+ residual)
    activations + the transformed weights, over the 29 3x3/stride-1 convolutions (stage: h, c, convs, with residual)."""
    tot = 0.0
    for h, c, convs, with_res in ((56, 64, 6, 3), (28, 128, 7, 4), (14, 256, 11, 6), (7, 512, 5, 3)):
        act = n_img * h * h * c * 4.0
        tot += convs * (2 * act + 6 * c * 3 * c * 4.0) + with_res * act
    return tot / 29


root, sub, out = sys.argv[1], sys.argv[2], sys.argv[3]
graphs = int(sys.argv[4]) if len(sys.argv) > 4 else 32
commit = sys.argv[5] if len(sys.argv) > 5 else None
f, nf = mean_counter(f"{root}/pmc_fetch/p_counter_collection.csv", sub, "FETCH_SIZE")
w, nw = mean_counter(f"{root}/pmc_write/p_counter_collection.csv", sub, "WRITE_SIZE")
res = {"kernel": sub, "launches_sampled": nf, "FETCH_SIZE_avg": f, "WRITE_SIZE_avg": w,
       "read_bytes_per_launch": 2.0 * f * 1024, "write_bytes_per_launch": w * 1024,
       "traffic_bytes_per_launch": 2.0 * f * 1024 + w * 1024,
       "correction": "bytes = 1024 * (2 * FETCH_SIZE + WRITE_SIZE): FETCH_SIZE counts 64 B per 128-B request on gfx950 "
                     "(MI355X_MICROARCH.md, HBM section); separate --pmc passes for the two counters",
       "command": "rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> -- python3 bench.py --steps 3 --warmup 1 "
                  "--cpu-baseline-seconds 0 --streams 1 --no-kernel-timing"}
# optional third pass: matrix-pipe occupancy (pmc_mfma: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_F32)
import os
mp = f"{root}/pmc_mfma/p_counter_collection.csv"
if os.path.exists(mp):
    busy, _ = mean_counter(mp, sub, "SQ_VALU_MFMA_BUSY_CYCLES")
    gui, _ = mean_counter(mp, sub, "GRBM_GUI_ACTIVE")
    cu, _ = mean_counter(mp, sub, "SQ_BUSY_CU_CYCLES")
    insts, _ = mean_counter(mp, sub, "SQ_INSTS_VALU_MFMA_F32")
    dur = []
    for r in csv.DictReader(open(f"{root}/pmc_mfma/p_kernel_trace.csv")):
        if sub in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
    cycles = gui / 8.0                                   # GRBM_GUI_ACTIVE is summed over the 8 XCDs
    res.update({"SQ_INSTS_VALU_MFMA_F32_avg": insts, "executed_mfma_gflop_per_launch": round(insts * 4096 / 1e9, 2),
                "SQ_VALU_MFMA_BUSY_CYCLES_avg": busy, "GRBM_GUI_ACTIVE_avg_sum_over_8_xcd": gui,
                "mfma_busy_frac": round(busy / (1024 * cycles), 4), "cu_busy_frac": round(cu / (256 * cycles), 4),
                "shader_clock_ghz": round(cycles / (sum(dur) / len(dur)) / 1e9, 3),
                "mfma_pass": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE "
                             "SQ_INSTS_VALU_MFMA_F32 (same command); busy = BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE/8)"})
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from relpose_gnn_amd.build import WINOGRAD_SOURCES, source_digest  # noqa: E402
alg = wino_algorithmic_bytes(8 * graphs)
res.update({"graphs_per_step": graphs, "images_per_launch": 8 * graphs, "source_digest": source_digest(WINOGRAD_SOURCES), "digest_of": list(WINOGRAD_SOURCES),
            "git_commit": commit,
            "algorithmic_bytes_per_launch": round(alg), "traffic_over_algorithmic": round(res["traffic_bytes_per_launch"] / alg, 3)})
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
