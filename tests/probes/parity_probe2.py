#!/usr/bin/env python3
"""Error of the HIP forward vs fp32 and fp64 CPU oracles for one 8-node graph at a given image shape, with the Winograd
path on and off.  usage: parity_probe2.py H W"""
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

H, W = int(sys.argv[1]), int(sys.argv[2])
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 12
dev = torch.device("cuda:0")
D = 2048
m = PoseNetX_R2(resnet34(), droprate=0.0, pretrained=False, feat_dim=D, edge_feat_dim=D, node_dim=D, input_img_height=H,
                use_gnn=True, knn=-1, use_AP=True, gnn_recursion=2)
sd = S.synth_state_dict(S.posenet_r2_param_shapes(D, D, D), seed=1)
m.load_state_dict(sd)
m = m.to(dev).eval()
x = S.synth_images(8, H, W, seed=seed)
rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max())
st32, st64 = {}, {}
oa, orr, _ = O.posenet_forward(sd, x, O.fc_edge_index(8), H, 2, st32)
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
oa64, or64, _ = O.posenet_forward(sd64, x.double(), O.fc_edge_index(8), H, 2, st64)
print(f"shape {H}x{W}: cpu fp32 vs fp64: abs {rel(oa, oa64):.2e} rel {rel(orr, or64):.2e} feat {rel(st32['fc'], st64['fc']):.2e}")
for wino in (1, 0):
    ops.set_tuning(ops.TUNE_WINOGRAD, wino)
    a, r, _ = m(fc_batch(x, 8).to(dev))
    feat = m._enc.run(m.feature_extractor.state_dict, "", x.view(8, 3, H, W).to(dev))
    print(f"wino={wino}: vs fp32: abs {rel(a.cpu(), oa):.2e} rel {rel(r.cpu(), orr):.2e} | vs fp64: abs {rel(a.cpu(), oa64):.2e} "
          f"rel {rel(r.cpu(), or64):.2e} feat {rel(feat.cpu(), st64['fc']):.2e}")
