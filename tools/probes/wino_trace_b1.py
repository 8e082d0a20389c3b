#!/usr/bin/env python3
"""Timeline of the Winograd launches of a SINGLE 8-node graph (8 images; the reference's batch_size=1 loop) from the RPG_WINO_TRACE
build (tools/probes/wino_trace.sh with SCRIPT=wino_trace_b1.py): every launch is all split-K parts of a few K steps each.  Prints,
per ResNet layer shape: launch wall time (events, kernel + fix-up), the span of the workgroups' lives and the medians of their
prologue / K loop / epilogue phases (s_memtime ticks, 100 MHz)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from relpose_gnn_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.lib()
lib.rpg_wino_trace_set.argtypes = [C.c_void_p]
n = int(os.environ.get("N", "8"))
for split in (4, 8, 2):
    ops.set_tuning(ops.TUNE_WINO_SPLIT, split)
    for (h, c, res) in ((56, 64, True), (28, 128, True), (14, 256, True), (7, 512, True)):
        x = torch.randn(n, h, h, c, device=dev)
        wt = torch.randn(c, 3, 3, c, device=dev) * 0.05
        u = ops.wino43_transform_weights(wt)
        sc, sh = torch.ones(c, device=dev), torch.zeros(c, device=dev)
        r = torch.randn(n, h, h, c, device=dev) if res else None
        buf = torch.zeros(5 * 8192, dtype=torch.int64, device=dev)
        for _ in range(50):
            ops.conv3x3_wino43_bn_act_nhwc(x, u, sc, sh, r, relu=True)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            ops.conv3x3_wino43_bn_act_nhwc(x, u, sc, sh, r, relu=True)
        b.record()
        torch.cuda.synchronize()
        per = a.elapsed_time(b) * 1e3 / 20
        assert lib.rpg_wino_trace_set(buf.data_ptr()) == 0
        ops.conv3x3_wino43_bn_act_nhwc(x, u, sc, sh, r, relu=True)
        torch.cuda.synchronize()
        assert lib.rpg_wino_trace_set(None) == 0
        t = buf.cpu().view(-1, 5).numpy()
        t = t[t[:, 1] != 0]
        pro, main, epi = t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3]
        span = t[:, 4].max() - t[:, 1].min()
        starts = np.sort(t[:, 1]) - t[:, 1].min()
        print(f"split>={split} {h}x{h}x{c}: {len(t)} workgroups; {per:.1f} us per conv (20 back-to-back, kernel + fix-up); workgroup span "
              f"{span/100:.1f} us; medians prologue {np.median(pro)/100:.2f} K loop {np.median(main)/100:.2f} epilogue {np.median(epi)/100:.2f} us; "
              f"last entry at {starts[-1]/100:.2f} us; mean life {np.mean(t[:,4]-t[:,1])/100:.2f} us", flush=True)
