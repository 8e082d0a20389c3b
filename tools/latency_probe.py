#!/usr/bin/env python3
"""Single-graph latency (the reference's batch_size=1 loop, testing/test.py:192) of the fp32 forward at 224x224 and 256x341: the
8 images of the graph on 1 / 2 / 4 HIP streams (model.small_batch_streams) x the per-layer Winograd / direct rule
(RPG_TUNE_WINOGRAD = n: Winograd from n blocks of 64 tiles x 64 channels per layer up; at one graph the layers have 98 / 50 / 28 / 16
blocks at 224x224) x eager / replayed from a captured HIP graph.  One line per variant: median wall time per call with a host
synchronisation after each call, and per call when 40 calls are streamed.   usage: tools/latency_probe.py [--graphs 1,2]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relpose_gnn_amd.synth as S  # noqa: E402
from relpose_gnn_amd import ops  # noqa: E402
from relpose_gnn_amd.graph import fc_batch  # noqa: E402
from relpose_gnn_amd.graphed import GraphedForward  # noqa: E402
from relpose_gnn_amd.posenet import PoseNetX_R2  # noqa: E402
from relpose_gnn_amd.resnet import resnet34  # noqa: E402

dev = torch.device("cuda:0")
D = 2048
glist = [int(v) for v in next((sys.argv[i + 1] for i, a in enumerate(sys.argv) if a == "--graphs"), "1").split(",")]
for (h, w) in ((224, 224), (256, 341)):
    m = PoseNetX_R2(resnet34(), droprate=0.0, pretrained=False, feat_dim=D, edge_feat_dim=D, node_dim=D, input_img_height=h,
                    use_gnn=True, knn=-1, use_AP=True, gnn_recursion=2)
    m.load_state_dict(S.synth_state_dict(S.posenet_r2_param_shapes(D, D, D), seed=1))
    m = m.to(dev).eval()
    for graphs in glist:
        x = torch.randn((8 * graphs, 3 * h * w), device=dev)
        d = fc_batch(x, 8)
        for rule in (16, 32, 64, 128):
            ops.set_tuning(ops.TUNE_WINOGRAD, rule)
            for streams in (1, 2, 4):
                if graphs > 1 and streams > 1:
                    continue                      # batches of >= 2 graphs x hip_streams go by graphs, not by images
                m.small_batch_streams = streams
                for graphed in (False, True):
                    if graphed and (rule != 16 or streams != 1):
                        continue
                    fn = GraphedForward(m, d) if graphed else m
                    for _ in range(5):
                        fn(d)
                    torch.cuda.synchronize()
                    ts = []
                    for _ in range(40):
                        t0 = time.perf_counter()
                        fn(d)
                        torch.cuda.synchronize()
                        ts.append(time.perf_counter() - t0)
                    t0 = time.perf_counter()
                    for _ in range(40):
                        fn(d)
                    torch.cuda.synchronize()
                    st = (time.perf_counter() - t0) / 40
                    print(f"{h}x{w} graphs {graphs} winograd-from {rule:3d} blocks  image_streams {streams} {'graph-replay' if graphed else 'eager       '}: "
                          f"latency {1e3 * sorted(ts)[20]:.3f} ms  streamed {1e3 * st:.3f} ms/call", flush=True)
        ops.set_tuning(ops.TUNE_WINOGRAD, 1)
