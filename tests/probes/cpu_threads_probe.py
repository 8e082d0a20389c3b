#!/usr/bin/env python3
"""CPU oracle throughput vs torch thread count on this host (to report the best CPU baseline, not a handicapped one)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import relpose_gnn_amd.synth as S  # noqa: E402
from oracle import posenet_ref as O  # noqa: E402

sd = S.synth_state_dict(S.posenet_r2_param_shapes(), seed=1)
for g in (4, 16):
    x = S.synth_images(8 * g, 224, 224, seed=77)
    ei = O.batch_edge_index(8, g)
    for th in (16, 32, 64, 128, 256):
        torch.set_num_threads(th)
        O.posenet_forward(sd, x, ei, 224, 2)
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 4.0:
            O.posenet_forward(sd, x, ei, 224, 2)
            n += 1
        dt = time.perf_counter() - t0
        print(f"graphs/forward={g:3d} threads={th:3d}: {g * n / dt:6.2f} graphs/s", flush=True)
