#!/usr/bin/env python3
"""The reference's evaluation loop as written (testing/test.py:205-251) over relpose_gnn_amd.lookahead at the evaluation shape
(8 x 256x341, fp32), stand-alone: graphs/s and where the host time goes.  usage: tools/lookahead_probe.py [graphs]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relpose_gnn_amd.synth as S  # noqa: E402
from relpose_gnn_amd import evaluate as E  # noqa: E402
from relpose_gnn_amd.graph import Batch, Data, fc_edge_index  # noqa: E402
from relpose_gnn_amd.lookahead import lookahead  # noqa: E402
from relpose_gnn_amd.posenet import PoseNetX_R2  # noqa: E402
from relpose_gnn_amd.resnet import resnet34  # noqa: E402

dev = torch.device("cuda:0")
D, h, w, mb = 2048, 256, 341, 64
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
m = PoseNetX_R2(resnet34(), droprate=0.0, pretrained=False, feat_dim=D, edge_feat_dim=D, node_dim=D, input_img_height=h,
                use_gnn=True, knn=-1, use_AP=True, gnn_recursion=2)
m.load_state_dict(S.synth_state_dict(S.posenet_r2_param_shapes(D, D, D), seed=1))
m = m.to(dev).eval()
gen = torch.Generator().manual_seed(77)
ei8 = fc_edge_index(8)
pool = [(torch.randn((8, 3 * h * w), generator=gen), torch.randn((8, 6), generator=gen) * 0.3) for _ in range(64)]
items = [Batch.from_data_list([Data(x=px, edge_index=ei8, y=py)]) for px, py in pool]


class _Loader:
    batch_size = 1

    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def __iter__(self):
        return (items[i % len(items)] for i in range(self.n))


def ref_loop(n, timing=None):
    loader, wrapped = lookahead(_Loader(n), m, dev, micro_batch=mb)
    preds = []
    t_iter = t_body = 0.0
    it = iter(enumerate(loader))
    while True:
        t0 = time.perf_counter()
        try:
            batch_idx, data = next(it)
        except StopIteration:
            break
        t1 = time.perf_counter()
        output, output_R, edge_index = wrapped(data.to(dev))
        s = output.size()
        output_R = output_R.cpu().data.numpy().reshape((-1, s[-1]))
        target = data.y.to("cpu").numpy().reshape((-1, s[-1]))
        edges = edge_index.cpu().data.numpy()
        preds.append(E.query_pose(output_R, target, edges, np.zeros(3), np.ones(3), 0)[0])
        t2 = time.perf_counter()
        t_iter += t1 - t0
        t_body += t2 - t1
    if timing is not None:
        timing.update(next_s=t_iter, body_s=t_body)
    return np.stack(preds), wrapped


for streams in (2, 1):
    m.hip_streams = streams
    ref_loop(2 * mb)
    torch.cuda.synchronize()
    for rep in range(2):
        tm = {}
        t0 = time.perf_counter()
        preds, wrapped = ref_loop(n, tm)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"hip_streams {streams}: {n / dt:.1f} graphs/s  ({dt:.3f} s; loader next() {tm['next_s']:.3f} s, loop body {tm['body_s']:.3f} s; forwards {wrapped.forwards}, direct {wrapped.direct_calls})", flush=True)
# the same stream through the product loop, for comparison
graphs = [Data(x=pool[i % 64][0], edge_index=ei8, y=pool[i % 64][1]) for i in range(n)]
E.evaluate_stream(m, graphs[:128], dev, micro_batch=mb)
torch.cuda.synchronize()
t0 = time.perf_counter()
E.evaluate_stream(m, graphs, dev, micro_batch=mb)
torch.cuda.synchronize()
print(f"evaluate_stream: {n / (time.perf_counter() - t0):.1f} graphs/s")
