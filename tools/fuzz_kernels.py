#!/usr/bin/env python3
"""Differential fuzzing of the convolution kernels against each other on random shapes (GPU box, ~30 s):
  * the Winograd kernels (4-wave, 8-wave, persistent 8-wave), with and without the split-K tail, vs the direct implicit-GEMM kernel (400 shapes);
  * the buffer-load loaders of the tile engine vs the general loaders over random tile / K-step / stream-K choices
    (300 shapes, strides 1-3, paddings 0-3, 4-channel tap-loader cases included).
Prints every mismatch and a final count; exit code 1 if any.     python tools/fuzz_kernels.py [--cases-scale 1.0]"""
import argparse
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from relpose_gnn_amd import ops  # noqa: E402


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def fuzz_winograd(dev, cases):
    bad, rng = 0, random.Random(12345)
    for case in range(cases):
        cin, cout = 4 * rng.randint(1, 48), 4 * rng.randint(1, 80)
        if case % 2:
            cin = 16 * rng.randint(1, 12)         # whole channel blocks: eligible for the persistent kernel
        n, h, w = rng.randint(1, 40), rng.randint(1, 40), rng.randint(1, 60)
        if case % 8 == 7:
            n = rng.randint(60, 200)              # more 128-tile workgroups than CUs: several items per workgroup
        res, relu = rng.random() < 0.5, rng.random() < 0.5
        g = torch.Generator().manual_seed(case)
        x = torch.randn(n, h, w, cin, generator=g).to(dev)
        wt = (torch.randn(cout, 3, 3, cin, generator=g) * (1.0 / (cin * 9)) ** 0.5).to(dev)
        sc, sh = (torch.rand(cout, generator=g) + 0.5).to(dev), (torch.randn(cout, generator=g) * 0.2).to(dev)
        r = torch.randn(n, h, w, cout, generator=g).to(dev) if res else None
        ref = ops.conv2d_bn_act_nhwc(x, wt, sc, sh, r, stride=1, pad=1, relu=relu)
        u = ops.wino43_transform_weights(wt)
        for kern, persist in ((2, 1), (3, 0), (3, 2)):      # persist 2: the persistent 8-wave kernel wherever Cin % 16 == 0
            for split in (0, 1):
                ops.set_tuning(ops.TUNE_WINOGRAD, kern)
                ops.set_tuning(ops.TUNE_WINO_SPLIT, split)
                ops.set_tuning(ops.TUNE_WINO_PERSIST, persist)
                e = rel(ops.conv3x3_wino43_bn_act_nhwc(x, u, sc, sh, r, relu=relu), ref)
                if not e < 3e-5:
                    bad += 1
                    print("MISMATCH winograd", case, kern, persist, split, (n, h, w, cin, cout, res, relu), e, flush=True)
    ops.set_tuning(ops.TUNE_WINOGRAD, 1)
    ops.set_tuning(ops.TUNE_WINO_SPLIT, 1)
    ops.set_tuning(ops.TUNE_WINO_PERSIST, 1)
    return bad


def fuzz_loaders(dev, cases):
    bad, rng = 0, random.Random(777)
    for case in range(cases):
        if case % 4 == 3:
            cin, kh, kw = 4, rng.choice([1, 3, 5, 7]), rng.choice([4, 5, 6, 7])
        else:
            cin, kh, kw = 16 * rng.randint(1, 12), rng.choice([1, 2, 3, 4]), rng.choice([1, 2, 3])
        stride, pad = rng.choice([1, 2, 3]), rng.choice([0, 1, 2, 3])
        cout = 4 * rng.randint(1, 70)
        n, h, w = rng.randint(1, 30), rng.randint(max(kh - 2 * pad, 1), 40), rng.randint(max(kw - 2 * pad, 1), 40)
        g = torch.Generator().manual_seed(case)
        x = torch.randn(n, h, w, cin, generator=g).to(dev)
        wt = (torch.randn(cout, kh, kw, cin, generator=g) * (1.0 / (cin * kh * kw)) ** 0.5).to(dev)
        sc, sh = (torch.rand(cout, generator=g) + 0.5).to(dev), (torch.randn(cout, generator=g) * 0.2).to(dev)
        ho, wo = (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1
        r = torch.randn(n, ho, wo, cout, generator=g).to(dev) if rng.random() < 0.5 else None
        tile, bk, sk = rng.choice([-1, 0, 1, 2, 3]), rng.choice([0, 16, 32]), rng.choice([0, 1])
        ops.set_tuning(ops.TUNE_TILE, tile)
        ops.set_tuning(ops.TUNE_BK, bk)
        ops.set_tuning(ops.TUNE_STREAMK, sk)
        ops.set_tuning(ops.TUNE_FAST_LOADER, 0)
        ref = ops.conv2d_bn_act_nhwc(x, wt, sc, sh, r, stride=stride, pad=pad, relu=True)
        ops.set_tuning(ops.TUNE_FAST_LOADER, 1)
        e = rel(ops.conv2d_bn_act_nhwc(x, wt, sc, sh, r, stride=stride, pad=pad, relu=True), ref)
        if not e < 2e-5:
            bad += 1
            print("MISMATCH loaders", case, (n, h, w, cin, cout, kh, kw, stride, pad, tile, bk, sk), e, flush=True)
    for k, v in ((ops.TUNE_TILE, -1), (ops.TUNE_BK, 0), (ops.TUNE_STREAMK, 1), (ops.TUNE_FAST_LOADER, 1)):
        ops.set_tuning(k, v)
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases-scale", type=float, default=1.0)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    bad = fuzz_winograd(dev, int(400 * args.cases_scale)) + fuzz_loaders(dev, int(300 * args.cases_scale))
    print("fuzz done, mismatches:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
