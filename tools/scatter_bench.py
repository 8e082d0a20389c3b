#!/usr/bin/env python3
"""Isolated bandwidth of the scatter-mean kernel vs batch size (8-node FC graphs, D=2048).  The in-pipeline launch at 32
graphs moves only 16.8 MB (cache resident, launch-latency sized); this shows where the kernel itself saturates.
Algorithmic bytes per graph per call: 524,736."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from relpose_gnn_amd import ops  # noqa: E402
from relpose_gnn_amd.graph import fc_edge_index  # noqa: E402

dev = torch.device("cuda:0")
D = 2048
for B in (16, 32, 128, 512, 2048, 8192):
    n, e = 8 * B, 56 * B
    ei = torch.cat([fc_edge_index(8) + 8 * g for g in range(B)], 1).to(dev)
    gp = ops.graph_prepare(ei, n)
    msg = torch.randn(e, D, device=dev)
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)          # > L2 + MALL, to evict msg between runs
    ts = []
    for rep in range(8):
        flush.fill_(rep)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = ops.scatter_mean(msg, gp["rowptr"], gp["perm"], n)
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    cold = ts[len(ts) // 2]
    for _ in range(3):
        ops.scatter_mean(msg, gp["rowptr"], gp["perm"], n)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        ops.scatter_mean(msg, gp["rowptr"], gp["perm"], n)
    b.record()
    b.synchronize()
    warm = a.elapsed_time(b) / 10
    nbytes = 524736.0 * B
    print(f"graphs={B:5d} bytes={nbytes/1e6:8.1f} MB  cold {cold*1e3:8.1f} us = {nbytes/cold/1e6:7.1f} GB/s ({nbytes/cold/8e9*100:4.1f}% of 8 TB/s)"
          f"   back-to-back {warm*1e3:8.1f} us = {nbytes/warm/1e6:7.1f} GB/s", flush=True)
