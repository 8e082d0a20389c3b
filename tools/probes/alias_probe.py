#!/usr/bin/env python3
"""Does the relative placement of a convolution's input / residual / output buffers matter (HBM channel aliasing)?
Layer-2 Winograd conv (256 images, 28x28x128), buffers carved from one allocation at 2-MiB-aligned bases plus a skew."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from relpose_gnn_amd import ops, _lib as L
dev = torch.device("cuda:0")
n, h, w, c = 256, 28, 28, 128
numel = n * h * w * c
big = torch.empty(4 * numel + (64 << 20), device=dev)            # floats
base = (-(big.data_ptr()) % (2 << 20)) // 4                        # first 2-MiB-aligned element
wt = torch.randn(c, 3, 3, c, device=dev) * 0.03
u = ops.wino43_transform_weights(wt)
sc, sh = torch.ones(c, device=dev), torch.zeros(c, device=dev)
stride = ((numel * 4 + (2 << 20) - 1) // (2 << 20)) * (2 << 20) // 4      # next multiple of 2 MiB, in floats
for skew_kb in (0, 4, 16, 68, 260, 1028, 2052, 4100):
    sk = skew_kb * 1024 // 4
    bufs = [big[base + i * (stride + sk): base + i * (stride + sk) + numel].view(n, h, w, c) for i in range(3)]
    x, r, y = bufs
    x.normal_(); r.normal_()
    def run():
        rc = L.lib().rpg_conv3x3_wino43_bn_act_nhwc_f32(x.data_ptr(), u.data_ptr(), sc.data_ptr(), sh.data_ptr(), r.data_ptr(),
                                                        y.data_ptr(), n, h, w, c, c, 1, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    ts = []
    for _ in range(3): run()
    torch.cuda.synchronize()
    for _ in range(20):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort()
    print(f"skew {skew_kb:5d} KB between consecutive buffers (x, residual, y): median {ts[10]*1e3:.1f} us  best {ts[0]*1e3:.1f}")
