"""Round-6 probe: what would a patch-resident kernel for the bf16 3x3 / stride-2 convolutions return?  Times each strided convolution of
the encoder (im2col LDS-DMA kernel: every input pixel fetched 9/4 times from L2) against a STRIDE-1 convolution of the same GEMM shape
(same M, N, K: the output map as input, same channels) on the im2col kernel and on the patch kernel (every pixel fetched once per
32-channel chunk): the gap between the last two is what the patch form buys at this shape, the first against the second what the strided
gather itself costs.   python tools/probes/s2_bound.py [images]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from relpose_gnn_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512


def timeit(fn, reps=20, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


g = torch.Generator(device=dev).manual_seed(1)
for name, hi, cin, cout in (("l2.s2", 56, 64, 128), ("l3.s2", 28, 128, 256), ("l4.s2", 14, 256, 512)):
    ho = hi // 2
    x2 = torch.randn((n, hi, hi, cin), generator=g, device=dev).bfloat16()
    x1 = torch.randn((n, ho, ho, cin), generator=g, device=dev).bfloat16()
    w = (torch.randn((cout, 3, 3, cin), generator=g, device=dev) * (2.0 / (9 * cin)) ** 0.5).bfloat16()
    sc, sh = torch.rand(cout, generator=g, device=dev) + 0.5, torch.randn(cout, generator=g, device=dev) * 0.1
    fl = 2.0 * n * ho * ho * cout * 9 * cin
    for rep in range(2):
        ops.set_tuning(ops.TUNE_BF16_PATCH, 1)
        t_s2 = timeit(lambda: ops.conv2d_bn_act_nhwc_bf16(x2, w, sc, sh, None, stride=2, pad=1, relu=True))
        ops.set_tuning(ops.TUNE_BF16_PATCH, 0)
        t_im = timeit(lambda: ops.conv2d_bn_act_nhwc_bf16(x1, w, sc, sh, None, stride=1, pad=1, relu=True))
        ops.set_tuning(ops.TUNE_BF16_PATCH, 2)
        t_pa = timeit(lambda: ops.conv2d_bn_act_nhwc_bf16(x1, w, sc, sh, None, stride=1, pad=1, relu=True))
        ops.set_tuning(ops.TUNE_BF16_PATCH, 1)
        print(f"{name} images={n} M={n * ho * ho} N={cout} K={9 * cin}: stride 2 (im2col) {t_s2:6.1f} us ({fl / t_s2 / 1e6:6.1f} TF) | same GEMM at stride 1: "
              f"im2col {t_im:6.1f} us, patch {t_pa:6.1f} us ({fl / t_pa / 1e6:6.1f} TF)", flush=True)
