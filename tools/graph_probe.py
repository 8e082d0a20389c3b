#!/usr/bin/env python3
"""Latency of one forward at small batch sizes, eager vs replayed from a captured HIP graph."""
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


def bench(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for B in (1, 2, 4, 8, 32):
    d = fc_batch(torch.randn(8 * B, 3 * H * H, device=dev), 8)
    eager = bench(lambda: m(d))
    out_e = m(d)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    try:
        with torch.cuda.stream(s):
            for _ in range(2):
                m(d)
            with torch.cuda.graph(g, stream=s):
                out = m(d)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        graphed = bench(lambda: g.replay())
        g.replay()
        torch.cuda.synchronize()
        ok = torch.allclose(out[1], out_e[1], rtol=1e-4, atol=1e-5)
        print(f"B={B:3d}: eager {eager*1e3:7.3f} ms  graph {graphed*1e3:7.3f} ms  ({B/eager:7.1f} -> {B/graphed:7.1f} graphs/s) same={ok}", flush=True)
    except Exception as e:  # noqa: BLE001
        print(f"B={B:3d}: eager {eager*1e3:7.3f} ms  capture failed: {type(e).__name__}: {str(e)[:200]}", flush=True)
