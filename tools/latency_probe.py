#!/usr/bin/env python3
"""Single-graph latency (the reference's batch_size=1 loop, testing/test.py:192) of the fp32 forward: eager vs replayed from a
captured HIP graph (graphed.GraphedForward), with the 8 images of the graph on 1 / 2 / 4 HIP streams (model.small_batch_streams).
Prints one line per variant: median wall time per call with a host synchronisation after each call, and per call streamed."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relpose_gnn_amd.synth as S  # noqa: E402
from relpose_gnn_amd.graph import fc_batch  # noqa: E402
from relpose_gnn_amd.graphed import GraphedForward  # noqa: E402
from relpose_gnn_amd.posenet import PoseNetX_R2  # noqa: E402
from relpose_gnn_amd.resnet import resnet34  # noqa: E402

dev = torch.device("cuda:0")
D = 2048
m = PoseNetX_R2(resnet34(), droprate=0.0, pretrained=False, feat_dim=D, edge_feat_dim=D, node_dim=D, input_img_height=224,
                use_gnn=True, knn=-1, use_AP=True, gnn_recursion=2)
m.load_state_dict(S.synth_state_dict(S.posenet_r2_param_shapes(D, D, D), seed=1))
m = m.to(dev).eval()
for graphs in (1, 2, 4):
    x = torch.randn((8 * graphs, 3 * 224 * 224), device=dev)
    d = fc_batch(x, 8)
    for streams in (1, 2, 4):
        m.small_batch_streams = streams
        for graphed in (False, True):
            if graphs > 1 and streams > 1:
                continue                      # batches of >= 2 graphs x hip_streams go by graphs, not by images
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
            print(f"graphs {graphs} image_streams {streams} {'graph-replay' if graphed else 'eager       '}: "
                  f"latency {1e3 * sorted(ts)[20]:.3f} ms  streamed {1e3 * st:.3f} ms/call", flush=True)
