#!/usr/bin/env python3
"""fp32 noise floor over many seeds, per fp32 stem kernel (round 6): HIP vs float64 oracle against CPU fp32 oracle vs float64, two
graphs per seed, 224x224 and 256x341.  usage: tools/probes/noise_floor_seeds.py [seeds]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import relpose_gnn_amd.synth as S  # noqa: E402
from conftest import rel_err  # noqa: E402
from oracle import posenet_ref as O  # noqa: E402
from relpose_gnn_amd import ops  # noqa: E402
from relpose_gnn_amd.graph import fc_batch  # noqa: E402
from relpose_gnn_amd.posenet import PoseNetX_R2  # noqa: E402
from relpose_gnn_amd.resnet import resnet34  # noqa: E402

dev = torch.device("cuda:0")
D = 2048
sd = S.synth_state_dict(S.posenet_r2_param_shapes(D, D, D), seed=1)
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.set_num_threads(32)
for (h, w) in ((224, 224), (256, 341)):
    m = PoseNetX_R2(resnet34(), droprate=0.0, pretrained=False, feat_dim=D, edge_feat_dim=D, node_dim=D, input_img_height=h,
                    use_gnn=True, knn=-1, use_AP=True, gnn_recursion=2)
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    sd64 = {k: v.double() for k, v in sd.items()}
    for seed in range(11, 11 + nseeds):
        x = torch.randn((16, 3 * h * w), generator=torch.Generator().manual_seed(seed))
        d = fc_batch(x, 8)
        oa, orr, _ = O.posenet_forward(sd, x, d.edge_index, h, 2)
        oa64, or64, _ = O.posenet_forward(sd64, x.double(), d.edge_index, h, 2)
        cpu_a, cpu_r = rel_err(oa, oa64), rel_err(orr, or64)
        row = f"{h}x{w} seed {seed}: cpu32-vs-64 abs {cpu_a:.2e} rel {cpu_r:.2e} |"
        for name, val in (("tile", 1), ("strips", 129)):
            ops.set_tuning(ops.TUNE_FUSED_STEM, val)
            a, r, _ = m(d.to(dev))
            ha, hr = rel_err(a.cpu(), oa64), rel_err(r.cpu(), or64)
            row += f" {name}: abs {ha:.2e} (x{ha / max(cpu_a, 3e-6):.2f}) rel {hr:.2e} (x{hr / max(cpu_r, 3e-6):.2f}) vs32 {rel_err(a.cpu(), oa):.2e} |"
        print(row, flush=True)
ops.set_tuning(ops.TUNE_FUSED_STEM, 1)
