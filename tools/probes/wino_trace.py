#!/usr/bin/env python3
"""Timeline of the 8-wave Winograd kernels (PERSIST=1: wino43_conv8p_kernel, per item; PERSIST=0: wino43_conv8_kernel) per CU from the RPG_WINO_TRACE build (tools/probes/wino_trace.sh): for every
workgroup HW_ID and s_memtime at entry / after the prologue / after the K loop / at exit.  Prints phase medians and, per
CU, the gaps between one workgroup's exit and the next one's entry (dispatch latency the in-kernel timers cannot see)."""
import ctypes as C
import os
import sys
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from relpose_gnn_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.lib()
lib.rpg_wino_trace_set.argtypes = [C.c_void_p]
ops.set_tuning(ops.TUNE_WINO_SPLIT, 0)
ops.set_tuning(ops.TUNE_WINOGRAD, 3)
ops.set_tuning(ops.TUNE_WINO_PERSIST, int(os.environ.get("PERSIST", "1")))
for (h, c, res) in ((56, 64, True), (28, 128, True), (14, 256, True), (7, 512, True)):
    n = 256
    x = torch.randn(n, h, h, c, device=dev)
    wt = torch.randn(c, 3, 3, c, device=dev) * 0.05
    u = ops.wino43_transform_weights(wt)
    sc, sh = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    r = torch.randn(n, h, h, c, device=dev) if res else None
    tiles = (n * h * ((h + 3) // 4) + 127) // 128 * ((c + 63) // 64)
    buf = torch.zeros(5 * tiles, dtype=torch.int64, device=dev)
    for _ in range(150):                 # clock ramp
        ops.conv3x3_wino43_bn_act_nhwc(x, u, sc, sh, r, relu=True)
    torch.cuda.synchronize()
    assert lib.rpg_wino_trace_set(buf.data_ptr()) == 0
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    ops.conv3x3_wino43_bn_act_nhwc(x, u, sc, sh, r, relu=True)
    b.record()
    torch.cuda.synchronize()
    assert lib.rpg_wino_trace_set(None) == 0
    t = buf.cpu().view(tiles, 5).numpy()
    hw = t[:, 0]
    cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 0x7) << 5)      # cu_id | sh_id | se_id (xcc is not in HW_ID)
    pro, main, epi = t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3]
    import numpy as np
    span = t[:, 4].max() - t[:, 1].min()
    print(f"== {h}x{h}x{c} res={res}: {tiles} workgroups, kernel {a.elapsed_time(b)*1e3:.1f} us, span {span} ticks (s_memtime, 100 MHz => {span/100:.1f} us)")
    print(f"   prologue {np.median(pro):.0f}  main {np.median(main):.0f}  epilogue {np.median(epi):.0f}  (ticks; p10/p90 main {np.percentile(main,10):.0f}/{np.percentile(main,90):.0f})")
    order = np.argsort(t[:, 1])
    starts = np.sort(t[:, 1]) - t[:, 1].min()
    print(f"   entries: first 256 within {starts[min(255, tiles-1)]} ticks; entry times of workgroups 256, 512, 1024: "
          f"{starts[min(256, tiles-1)]}, {starts[min(512, tiles-1)]}, {starts[min(1024, tiles-1)]}")
    print(f"   mean workgroup life {np.mean(t[:,4]-t[:,1]):.0f} ticks; sum of lives / 256 CUs = {np.sum(t[:,4]-t[:,1])/256:.0f} ticks vs span {span}")
    if int(os.environ.get("PERSIST", "1")) and tiles > 256:
        for b in (0, 101):               # the items of one persistent workgroup, relative to its first timestamp
            its = t[b::256]
            base = its[0, 1]
            print(f"   workgroup {b}: " + "  ".join(f"[{r[1]-base} {r[2]-base} {r[3]-base} {r[4]-base}]" for r in its))
