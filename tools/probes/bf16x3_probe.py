#!/usr/bin/env python3
"""VERDICT r2 item 7 (experiment, reported separately -- `value` stays native f32): fp32 products on the bf16 matrix pipe by
exact 3-way bf16 splitting of both operands (x = x0 + x1 + x2, each bf16; 8 + 8 + 8 mantissa bits) with fp32 accumulation,
on ONE layer-3 convolution (256 -> 256 channels, 14 x 14, 3x3 / stride 1).

What is measured here: (1) the ACCURACY of the 6-product form (x0w0 + x0w1 + x1w0 + x0w2 + x1w1 + x2w0; the three dropped
products are below 2^-24 relative) and of the full 9-product form, against an fp64 reference, next to the f32 Winograd
kernel and the direct f32 MFMA kernel on the same data; (2) the TIME of the pieces it would be built from: the bf16 patch
kernel with fp32 output, run once per product (an unfused emulation: 6 launches + 5 additions), and the f32 Winograd kernel
it would have to beat.  A fused kernel would issue 6 bf16 MFMAs per (A, B) fragment pair = 6/16 of the direct f32 MFMA time
at equal matrix-pipe efficiency; the unfused emulation is an upper bound on its time, not a product path.

    python tools/probes/bf16x3_probe.py [--nimg 256]
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from relpose_gnn_amd import ops  # noqa: E402


def split3(t):
    a = t.bfloat16()
    r = t - a.float()
    b = r.bfloat16()
    c = (r - b.float()).bfloat16()
    return a, b, c


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2] * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nimg", type=int, default=256)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    n, h, c = args.nimg, 14, 256
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, h, h, c, generator=g)
    wt = torch.randn(c, 3, 3, c, generator=g) * (2.0 / (9 * c)) ** 0.5
    one, zero = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    xd, wd = x.to(dev), wt.to(dev)
    # fp64 reference on a subset of images (CPU)
    ns = min(n, 8)
    ref = F.conv2d(x[:ns].permute(0, 3, 1, 2).double(), wt.permute(0, 3, 1, 2).double(), padding=1).permute(0, 2, 3, 1)
    err = lambda y: float((y[:ns].double().cpu() - ref).abs().max() / ref.abs().max())

    xs, ws = split3(xd), split3(wd)
    conv = lambda a, b: ops.conv2d_bn_act_nhwc_bf16(a, b, one, zero, None, stride=1, pad=1, relu=False, out_f32=True)
    pairs6 = [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)]
    pairs9 = pairs6 + [(1, 2), (2, 1), (2, 2)]

    def run(pairs):
        acc = None
        for i, j in reversed(pairs):          # small terms first
            y = conv(xs[i], ws[j])
            acc = y if acc is None else acc + y
        return acc
    y6, y9 = run(pairs6), run(pairs9)
    y1 = conv(xs[0], ws[0])
    u = ops.wino43_transform_weights(wd)
    yw = ops.conv3x3_wino43_bn_act_nhwc(xd, u, one, zero, None, relu=False)
    ydir = ops.conv2d_bn_act_nhwc(xd, wd, one, zero, None, stride=1, pad=1, relu=False)
    print(f"layer-3 convolution, {n} images: max-norm relative error vs fp64 ({ns} images)")
    print(f"  bf16 x bf16 (1 product)           {err(y1):.3e}")
    print(f"  3-way split, 6 products (fp32 acc) {err(y6):.3e}")
    print(f"  3-way split, 9 products (fp32 acc) {err(y9):.3e}")
    print(f"  f32 Winograd F(4,3) kernel         {err(yw):.3e}")
    print(f"  f32 direct MFMA kernel             {err(ydir):.3e}")
    t1 = timeit(lambda: conv(xs[0], ws[0]))
    t6 = timeit(lambda: run(pairs6))
    tw = timeit(lambda: ops.conv3x3_wino43_bn_act_nhwc(xd, u, one, zero, None, relu=False))
    td = timeit(lambda: ops.conv2d_bn_act_nhwc(xd, wd, one, zero, None, stride=1, pad=1, relu=False))
    print(f"time: one bf16 product (patch kernel, fp32 out) {t1:.1f} us; 6 products unfused + 5 adds {t6:.1f} us; "
          f"f32 Winograd {tw:.1f} us; f32 direct {td:.1f} us")
    print(f"  a fused 6-product kernel at the bf16 kernel's matrix-pipe efficiency would take about {6 * t1 * 0.75:.0f}-{6 * t1:.0f} us "
          "(operand traffic shared between the products) -- against the Winograd kernel's time above")


if __name__ == "__main__":
    main()
