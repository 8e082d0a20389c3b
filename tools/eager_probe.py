#!/usr/bin/env python3
"""Why is eager 2-stream B=32 slower in graph_probe.py than in bench.py?  Vary what precedes the timing loop."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relpose_gnn_amd.synth as S  # noqa: E402
from relpose_gnn_amd.graph import fc_batch  # noqa: E402
from relpose_gnn_amd.posenet import PoseNetX_R2  # noqa: E402
from relpose_gnn_amd.resnet import resnet34  # noqa: E402

dev = torch.device("cuda:0")
D, H = 2048, 224
m = PoseNetX_R2(resnet34(), droprate=0.0, pretrained=False, feat_dim=D, edge_feat_dim=D, node_dim=D, input_img_height=H,
                use_gnn=True, knn=-1, use_AP=True, gnn_recursion=2)
m.load_state_dict(S.synth_state_dict(S.posenet_r2_param_shapes(D, D, D), seed=1))
m = m.to(dev).eval()


def bench(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


d32 = fc_batch(torch.randn(8 * 32, 3 * H * H, device=dev), 8)
print("fresh model, B=32 eager:", round(bench(lambda: m(d32)), 3), "ms")
d4 = fc_batch(torch.randn(8 * 4, 3 * H * H, device=dev), 8)
print("B=4 eager:", round(bench(lambda: m(d4)), 3), "ms")
print("B=32 eager again (after a different shape ran):", round(bench(lambda: m(d32)), 3), "ms")
d1 = fc_batch(torch.randn(8, 3 * H * H, device=dev), 8)
print("B=1 eager:", round(bench(lambda: m(d1)), 3), "ms")
print("B=32 eager again:", round(bench(lambda: m(d32)), 3), "ms")
t = time.perf_counter()
for _ in range(5):
    m(d32)
    torch.cuda.synchronize()
print("B=32 eager, sync every step:", round((time.perf_counter() - t) / 5 * 1e3, 3), "ms")
