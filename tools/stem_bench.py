#!/usr/bin/env python3
"""Time the fused stem kernel (conv7x7/2 + BN + ReLU + maxpool) at the bench shape: 256 images 224x224."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from relpose_gnn_amd import ops  # noqa: E402
from relpose_gnn_amd.params import pack_stem_pairs  # noqa: E402

dev = torch.device("cuda:0")
n, h, w = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 224, 224
x = torch.randn(n, 3, h, w, device=dev)
wt = torch.randn(64, 3, 7, 7) * 0.1
wp = pack_stem_pairs(wt, torch.ones(64)).to(dev)
sh = torch.zeros(64, device=dev)
for _ in range(3):
    ops.stem_conv_bn_relu_maxpool(x, wp, sh)
torch.cuda.synchronize()
ts = []
for _ in range(20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); ops.stem_conv_bn_relu_maxpool(x, wp, sh); b.record(); b.synchronize()
    ts.append(a.elapsed_time(b))
ts.sort()
flop = 2.0 * n * 112 * 112 * 64 * 147
print(f"stem n={n}: median {ts[10]*1e3:.1f} us  best {ts[0]*1e3:.1f} us   {flop/ts[10]/1e9:.1f} TFLOP/s algorithmic")
