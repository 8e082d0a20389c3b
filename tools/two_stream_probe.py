#!/usr/bin/env python3
"""Does running the batch as two concurrent half-batches on two HIP streams recover the tile-quantisation tails?"""
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
sd = S.synth_state_dict(S.posenet_r2_param_shapes(D, D, D), seed=1)


def mk():
    m = PoseNetX_R2(resnet34(), droprate=0.0, pretrained=False, feat_dim=D, edge_feat_dim=D, node_dim=D,
                    input_img_height=H, use_gnn=True, knn=-1, use_AP=True, gnn_recursion=2)
    m.load_state_dict(sd)
    return m.to(dev).eval()


def run(parts, B, steps=10):
    models = [mk() for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    datas = [fc_batch(torch.randn(8 * B // parts, 3 * H * H, device=dev), 8) for _ in range(parts)]

    def step():
        for m, s, d in zip(models, streams, datas):
            with torch.cuda.stream(s):
                m(d)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"B={B} parts={parts}: {dt*1e3:.2f} ms/step  {B/dt:.0f} graphs/s", flush=True)


for B, parts in ((32, 1), (32, 2), (32, 4), (64, 1), (64, 2), (64, 4)):
    run(parts, B)
