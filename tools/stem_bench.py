#!/usr/bin/env python3
"""Time the fused stem kernels (conv7x7/2 + BN + ReLU + maxpool): the fp32 one at 256 images 224x224 and the bf16 ones
(strip-march kernel of round 6 in its variants, tile kernel of rounds 3-5) at 512 images.

    python tools/stem_bench.py [images] [--bf16] [--shape HxW]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from relpose_gnn_amd import ops  # noqa: E402
from relpose_gnn_amd.params import pack_stem_bf16, pack_stem_pairs  # noqa: E402

dev = torch.device("cuda:0")
_vals = {sys.argv[i + 1] for i, a in enumerate(sys.argv) if a in ("--shape", "--variants") and i + 1 < len(sys.argv)}
args = [a for a in sys.argv[1:] if not a.startswith("--") and a not in _vals]
shape = next((sys.argv[i + 1] for i, a in enumerate(sys.argv) if a == "--shape"), "224x224")
h, w = (int(v) for v in shape.split("x"))
bf16 = "--bf16" in sys.argv
n = int(args[0]) if args else (512 if bf16 else 256)
x = torch.randn(n, 3, h, w, device=dev)
wt = torch.randn(64, 3, 7, 7) * 0.1
hc, wc = (h - 1) // 2 + 1, (w - 1) // 2 + 1
flop = 2.0 * n * hc * wc * 64 * 147


def timeit(fn, reps=20):
    for _ in range(40):            # (clock ramp: a cold GPU runs these kernels up to 25 % slower for the first dozen launches)
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


if not bf16:
    wp = pack_stem_pairs(wt, torch.ones(64)).to(dev)
    sh = torch.zeros(64, device=dev)
    for name, val in (("tile kernel (default)", 1), ("strips", 129), ("strips, bands of 7", 129 + (7 << 8)), ("strips, bands of 14", 129 + (14 << 8)),
                      ("strips, bands of 28", 129 + (28 << 8)), ("tile kernel (default)", 1), ("strips", 129)):
        ops.set_tuning(ops.TUNE_FUSED_STEM, val)
        med, best = timeit(lambda: ops.stem_conv_bn_relu_maxpool(x, wp, sh))
        print(f"fp32 stem n={n} {h}x{w} {name:24s}: median {med*1e3:7.1f} us  best {best*1e3:7.1f} us   {flop/med/1e9:6.1f} TFLOP/s algorithmic", flush=True)
    ops.set_tuning(ops.TUNE_FUSED_STEM, 1)
else:
    wp = pack_stem_bf16(wt).to(dev)
    sc, sh = torch.rand(64, device=dev) + 0.5, torch.zeros(64, device=dev)
    xb = x.bfloat16()
    only = next((sys.argv[i + 1] for i, a in enumerate(sys.argv) if a == "--variants"), "")
    for name, val in (("tile kernel (r3-r5)", 3), ("strips, default (both halves per wave)", 1), ("strips, one half per wave, 3 waves/SIMD", 33),
                      ("strips, default, bands of 7", 1 + (7 << 8)), ("strips, default, bands of 14", 1 + (14 << 8)), ("strips, default, bands of 28", 1 + (28 << 8)),
                      ("strips, default, bands of 56", 1 + (56 << 8))):
        if only and only not in name:
            continue
        ops.set_tuning(ops.TUNE_FUSED_STEM, val)
        for xin, tag in ((x, "fp32 in"), (xb, "bf16 in")):
            med, best = timeit(lambda: ops.stem_conv_bn_relu_maxpool_bf16(xin, wp, sc, sh))
            print(f"bf16 stem n={n} {h}x{w} {name:42s} {tag}: median {med*1e3:7.1f} us  best {best*1e3:7.1f} us   {flop/med/1e9:7.1f} TFLOP/s algorithmic", flush=True)
    ops.set_tuning(ops.TUNE_FUSED_STEM, 1)
