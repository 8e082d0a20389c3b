#!/usr/bin/env python3
"""A/B of RPG_TUNE_FIXUP_PRIO (round 6): the headline step (32 graphs x 8 x 224x224, fp32, two streams) with the
fix-up launches of split tiles on the launch stream (0) / on a high-priority companion stream (1): bitwise equality of the poses,
then ms per step, alternating, 5 rounds of 20 steps each."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relpose_gnn_amd.synth as S  # noqa: E402
from relpose_gnn_amd import ops  # noqa: E402
from relpose_gnn_amd.graph import fc_batch  # noqa: E402
from relpose_gnn_amd.posenet import PoseNetX_R2  # noqa: E402
from relpose_gnn_amd.resnet import resnet34  # noqa: E402

dev = torch.device("cuda:0")
D = 2048
m = PoseNetX_R2(resnet34(), droprate=0.0, pretrained=False, feat_dim=D, edge_feat_dim=D, node_dim=D, input_img_height=224,
                use_gnn=True, knn=-1, use_AP=True, gnn_recursion=2)
m.load_state_dict(S.synth_state_dict(S.posenet_r2_param_shapes(D, D, D), seed=1))
m = m.to(dev).eval()
streams = int(sys.argv[1]) if len(sys.argv) > 1 else 2
m.hip_streams = streams
x = torch.randn((256, 3 * 224 * 224), generator=torch.Generator(device=dev).manual_seed(1234), device=dev)
d = fc_batch(x, 8)
outs = {}
for mode in (0, 1):
    ops.set_tuning(ops.TUNE_FIXUP_PRIO, mode)
    for _ in range(3):
        a, r, _ = m(d)
    torch.cuda.synchronize()
    outs[mode] = (a.clone(), r.clone())
print("bitwise equal:", torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), flush=True)
res = {0: [], 1: []}
for rnd in range(5):
    for mode in (0, 1):
        ops.set_tuning(ops.TUNE_FIXUP_PRIO, mode)
        for _ in range(3):
            m(d)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            m(d)
        torch.cuda.synchronize()
        res[mode].append(1e3 * (time.perf_counter() - t0) / 20)
ops.set_tuning(ops.TUNE_FIXUP_PRIO, 0)
for mode in (0, 1):
    v = sorted(res[mode])
    print(f"streams {streams} fixup_prio {mode}: ms/step median {v[2]:.3f}  min {v[0]:.3f}  max {v[-1]:.3f}   all {[round(t, 3) for t in res[mode]]}", flush=True)
