#!/usr/bin/env python3
"""Error of the HIP forward vs the CPU oracle for one graph evaluated inside a 32-graph batch, under the tile-engine
variants (stream-K on/off, K-step).  Settles whether a variant adds more than summation-order noise."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import relpose_gnn_amd.synth as S  # noqa: E402
from oracle import posenet_ref as O  # noqa: E402
from relpose_gnn_amd import ops  # noqa: E402
from relpose_gnn_amd.graph import fc_batch  # noqa: E402
from relpose_gnn_amd.posenet import PoseNetX_R2  # noqa: E402
from relpose_gnn_amd.resnet import resnet34  # noqa: E402

dev = torch.device("cuda:0")
D, H = 2048, 64
m = PoseNetX_R2(resnet34(), droprate=0.0, pretrained=False, feat_dim=D, edge_feat_dim=D, node_dim=D, input_img_height=H,
                use_gnn=True, knn=-1, use_AP=True, gnn_recursion=2)
sd = S.synth_state_dict(S.posenet_r2_param_shapes(D, D, D), seed=1)
m.load_state_dict(sd)
m = m.to(dev).eval()
x = S.synth_images(8 * 32, H, H, seed=8)
rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max())
stages = {}
oa, orr, _ = O.posenet_forward(sd, x[40:48], O.fc_edge_index(8), H, 2, stages)
# float64 oracle for the same graph: which of the two fp32 results is closer to the exact answer?
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
oa64, or64, _ = O.posenet_forward(sd64, x[40:48].double(), O.fc_edge_index(8), H, 2)
print(f"cpu fp32 oracle vs fp64 oracle: abs {rel(oa, oa64):.2e} rel {rel(orr, or64):.2e}")
for sk in (1, 0):
    for bk in (0, 16, 32):
        ops.set_tuning(ops.TUNE_STREAMK, sk)
        ops.set_tuning(ops.TUNE_BK, bk)
        a, r, _ = m(fc_batch(x, 8).to(dev))
        a1, r1, _ = m(fc_batch(x[40:48], 8).to(dev))
        print(f"sk={sk} bk={bk:2d}  in-batch vs fp32 oracle: abs {rel(a[40:48].cpu(), oa):.2e} rel {rel(r[280:336].cpu(), orr):.2e} | "
              f"vs fp64: abs {rel(a[40:48].cpu(), oa64):.2e} rel {rel(r[280:336].cpu(), or64):.2e} | alone vs fp64: abs {rel(a1.cpu(), oa64):.2e}")
