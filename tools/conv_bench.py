#!/usr/bin/env python3
"""Micro-benchmark of the implicit-GEMM conv / gathered GEMM kernels at the ResNet34 / GNN shapes of BASELINE configs[1]
(256 images, 1792 edges).  Prints TFLOP/s per shape (HIP-event timed, median of reps).  Usage on the GPU box:
    python tools/conv_bench.py [--reps 20] [--only l3]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from relpose_gnn_amd import ops  # noqa: E402

SHAPES = [  # name, n, h, w, cin, cout, k, stride, pad, residual
    ("stem", 256, 224, 224, 4, 64, 7, 2, 3, False),
    ("l1.c1", 256, 56, 56, 64, 64, 3, 1, 1, False),
    ("l1.c2", 256, 56, 56, 64, 64, 3, 1, 1, True),
    ("l2.c1", 256, 28, 28, 128, 128, 3, 1, 1, False),
    ("l2.c2", 256, 28, 28, 128, 128, 3, 1, 1, True),
    ("l2.ds", 256, 56, 56, 64, 128, 1, 2, 0, False),
    ("l2.s2", 256, 56, 56, 64, 128, 3, 2, 1, False),
    ("l3.s2", 256, 28, 28, 128, 256, 3, 2, 1, False),
    ("l3.ds", 256, 28, 28, 128, 256, 1, 2, 0, False),
    ("l4.s2", 256, 14, 14, 256, 512, 3, 2, 1, False),
    ("l4.ds", 256, 14, 14, 256, 512, 1, 2, 0, False),
    ("l3.c1", 256, 14, 14, 256, 256, 3, 1, 1, False),
    ("l3.c2", 256, 14, 14, 256, 256, 3, 1, 1, True),
    ("l4.c1", 256, 7, 7, 512, 512, 3, 1, 1, False),
    ("l4.c2", 256, 7, 7, 512, 512, 3, 1, 1, True),
]
LINEAR = [  # name, m, widths, n_out
    ("edge0", 1792, (2048, 2048, 2048), 2048),
    ("edge2", 1792, (2048,), 2048),
    ("gtp", 1792, (2048,), 768),
    ("attW", 1792, (256,), 2048),
    ("upd0", 256, (2048, 2048), 2048),
    ("fc", 256, (512,), 2048),
    # the per-NODE GEMMs of the split formulation (forward.hip node_gemm): M = nodes
    ("nd.projn", 256, (2048,), 4096),
    ("nd.node3", 256, (2048,), 6144),
    ("nd.upd2", 256, (2048,), 2048),
    ("nd.attW", 256, (256,), 2048),
    ("nd128.projn", 128, (2048,), 4096),
    ("nd128.node3", 128, (2048,), 6144),
    ("nd128.upd0", 128, (2048, 2048), 2048),
    ("nd128.upd2", 128, (2048,), 2048),
    # round 6: what a four-way split-K of the attention projections on exact-fit 112 x 64 tiles would cost --
    # the same 768 work items of 16 K steps each, emulated as ONE Linear of 4 x 1792 rows and K = 512 (then + a fix-up launch)
    ("gtp.splitk4_emulated", 4 * 1792, (512,), 768),
    ("gtp.splitk2_emulated", 2 * 1792, (1024,), 768),
    # ... and the algebraic merge (the projections as 768 more output columns of mlp.2's GEMM, weight product precomputed)
    ("gtp.merged_with_msg2", 1792, (2048,), 2816),
    ("gtp.msg2_alone", 1792, (2048,), 2048),
    ("gtp.m896", 896, (2048,), 768),
    ("gtp.m896_merged_with_msg2", 896, (2048,), 2816),
    ("gtp.m896_msg2_alone", 896, (2048,), 2048),
]


WARM = 0


def timeit(fn, reps):
    for _ in range(1 + WARM):      # --warm N: N untimed launches first (clock ramp: a cold GPU runs these kernels ~25 % slower)
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
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--bk", type=int, default=0)
    ap.add_argument("--epi", type=int, default=1)
    ap.add_argument("--tile", type=int, default=-1)
    ap.add_argument("--sk", type=int, default=1)
    ap.add_argument("--wino", type=int, default=1)
    ap.add_argument("--wino2d-ab", action="store_true", help="fp32 3x3/s1 shapes: time the 1-D F(4,3) and the nested F(4x2,3x3) kernels interleaved")
    ap.add_argument("--nimg", type=int, default=256, help="images in the batch (256 = 32 graphs)")
    ap.add_argument("--warm", type=int, default=0, help="untimed launches before the timed ones")
    ap.add_argument("--bf16", action="store_true", help="the bf16 convolution kernels (bf16 activations / weights) instead of fp32")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE", help="rpg_set_tuning(KEY, VALUE)")
    ap.add_argument("--block64", action="store_true", help="bf16: the fused 64-channel BasicBlock kernel (rpg_basicblock64_bf16) against its "
                    "two convolution launches on the same operands, interleaved (layer 1 of ResNet34: 56x56x64)")
    ap.add_argument("--block64-map", default="56x56", help="with --block64: the layer-1 map (56x56 at 224x224 images, 64x86 at 256x341)")
    ap.add_argument("--dma-sweep", default="", help="bf16 only: comma-separated configuration indices of the LDS-DMA kernel; every shape "
                    "is timed with the default dispatch and with each of them (RPG_TUNE_BF16_DMA = 10 + i) interleaved in one process")
    args = ap.parse_args()
    global WARM
    WARM = args.warm
    for kv in args.tune:
        k, v = kv.split("=")
        ops.set_tuning(int(k), int(v))
    dev = torch.device("cuda:0")
    ops.set_tuning(ops.TUNE_BK, args.bk)
    ops.set_tuning(ops.TUNE_EPILOGUE, args.epi)
    ops.set_tuning(ops.TUNE_TILE, args.tile)
    ops.set_tuning(ops.TUNE_STREAMK, args.sk)
    ops.set_tuning(ops.TUNE_WINOGRAD, args.wino)
    if os.environ.get("RPG_WS64"):
        ops.set_tuning(ops.TUNE_BF16_WS64, int(os.environ["RPG_WS64"]))
    print(f"# bk={args.bk} epi={args.epi} tile={args.tile} streamk={args.sk}", flush=True)
    if args.block64:
        h, w = (int(v) for v in args.block64_map.split("x"))
        n = args.nimg
        g = torch.Generator(device=dev).manual_seed(5)
        x = torch.randn((n, h, w, 64), generator=g, device=dev).bfloat16()
        w1 = (torch.randn((64, 3, 3, 64), generator=g, device=dev) * (2.0 / 576) ** 0.5).bfloat16()
        w2 = (torch.randn((64, 3, 3, 64), generator=g, device=dev) * (2.0 / 576) ** 0.5).bfloat16()
        s1, b1 = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.1
        s2, b2 = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.1

        def two():
            t = ops.conv2d_bn_act_nhwc_bf16(x, w1, s1, b1, None, stride=1, pad=1, relu=True)
            return ops.conv2d_bn_act_nhwc_bf16(t, w2, s2, b2, x, stride=1, pad=1, relu=True)
        fl = 2 * 2.0 * n * h * w * 64 * 576
        same = torch.equal(two(), ops.basicblock64_bf16(x, w1, s1, b1, w2, s2, b2))
        def fused(mode):
            ops.set_tuning(ops.TUNE_BF16_FUSE_BLOCK, mode)
            return ops.basicblock64_bf16(x, w1, s1, b1, w2, s2, b2)
        base = fused(1)
        for mode in (1, 3, 3, 5, 5):
            y = fused(mode)
            d = (y != base)
            if d.any():
                idx = d.nonzero()
                print(f"# mode {mode}: {int(d.sum())} elements differ from the first run; first {idx[0].tolist()} last {idx[-1].tolist()} "
                      f"rows {sorted(set(idx[:, 1].tolist()))[:12]} cols {sorted(set(idx[:, 2].tolist()))[:12]}", flush=True)
            else:
                print(f"# mode {mode}: identical to the first run", flush=True)
        same_ps = torch.equal(fused(3), base)
        for rep in range(3):
            ops.set_tuning(ops.TUNE_BF16_FUSE_BLOCK, 3)
            mp, _ = timeit(lambda: ops.basicblock64_bf16(x, w1, s1, b1, w2, s2, b2), args.reps)
            ops.set_tuning(ops.TUNE_BF16_FUSE_BLOCK, 5)
            mpp, _ = timeit(lambda: ops.basicblock64_bf16(x, w1, s1, b1, w2, s2, b2), args.reps)
            print(f"block64 images={n} map {h}x{w}  two-group {mpp*1e3:7.1f} us   ", end="")
            ops.set_tuning(ops.TUNE_BF16_FUSE_BLOCK, 1)
            m2, _ = timeit(two, args.reps)
            m1, _ = timeit(lambda: ops.basicblock64_bf16(x, w1, s1, b1, w2, s2, b2), args.reps)
            print(f"block64 images={n} map {h}x{w}  persistent {mp*1e3:7.1f} us (== one tile per workgroup: {same_ps})   ", end="")
            print(f"block64 images={n} map {h}x{w}  two launches {m2*1e3:7.1f} us ({fl/m2/1e9:6.1f} TF)   fused {m1*1e3:7.1f} us ({fl/m1/1e9:6.1f} TF)   "
                  f"ratio {m1/m2:.3f}   bit-identical {same}", flush=True)
        return
    for name, n, h, w, cin, cout, k, s, p, res in SHAPES:
        if args.only and args.only not in name:
            continue
        n = args.nimg
        x = torch.randn(n, h, w, cin, device=dev)
        wt = torch.randn(cout, k, k, cin, device=dev) * (2.0 / (cin * k * k)) ** 0.5
        sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.1
        ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
        r = torch.randn(n, ho, wo, cout, device=dev) if res else None
        if args.bf16:
            if name == "stem":
                from importlib import import_module
                P = import_module("relpose-gnn_amd.params")
                xin = torch.randn(n, 3, h, w, device=dev)
                wp = P.pack_stem_bf16(torch.randn(64, 3, 7, 7) * 0.1).to(dev)
                med, best = timeit(lambda: ops.stem_conv_bn_relu_maxpool_bf16(xin, wp, sc, sh), args.reps)
                fl = 2.0 * n * ho * wo * cout * 147
                print(f"fused bf16 stem  images={n}  {med*1e3:8.1f} us  {fl/med/1e9:7.1f} TF (best {fl/best/1e9:6.1f})", flush=True)
                continue
            xb, wb = x.bfloat16(), wt.bfloat16()
            rb = None if r is None else r.bfloat16()
            run = lambda: ops.conv2d_bn_act_nhwc_bf16(xb, wb, sc, sh, rb, stride=s, pad=p, relu=True)
            if args.dma_sweep:
                fl = 2.0 * n * ho * wo * cout * k * k * cin
                cells = []
                ops.set_tuning(ops.TUNE_BF16_PATCH, 0)
                for cfg in [-1] + [int(v) for v in args.dma_sweep.split(",") if v.strip()]:
                    ops.set_tuning(ops.TUNE_BF16_DMA, 0 if cfg < 0 else 10 + cfg)
                    med, best = timeit(run, args.reps)
                    cells.append(f"{'base' if cfg < 0 else 'c%d' % cfg}:{med*1e3:6.1f}us/{fl/med/1e9:5.0f}TF")
                ops.set_tuning(ops.TUNE_BF16_DMA, 1)
                med, best = timeit(run, args.reps)
                cells.append(f"auto:{med*1e3:6.1f}us/{fl/med/1e9:5.0f}TF")
                if k == 3 and s == 1:
                    for mode in (2, 3):
                        ops.set_tuning(ops.TUNE_BF16_PATCH, mode)
                        med, best = timeit(run, args.reps)
                        cells.append(f"patch{mode}:{med*1e3:6.1f}us/{fl/med/1e9:5.0f}TF")
                ops.set_tuning(ops.TUNE_BF16_PATCH, 1)
                print(f"conv {name:6s} M={n*ho*wo:8d} N={cout:4d} K={k*k*cin:5d}  " + "  ".join(cells), flush=True)
                continue
            med, best = timeit(run, args.reps)
        elif args.wino and k == 3 and s == 1 and args.wino2d_ab:
            # A/B of the two Winograd forms on the same operands, interleaved in one process: 1-D F(4,3) (RPG_TUNE_WINO2D = 0)
            # against the nested F(4x2, 3x3) (= 2), + their relative max-norm difference
            u = ops.wino43_transform_weights(wt)
            fl = 2.0 * n * ho * wo * cout * 9 * cin
            cells, outs = [], {}
            for mode in (0, 2, 0, 2):
                ops.set_tuning(ops.TUNE_WINO2D, mode)
                med, best = timeit(lambda: ops.conv3x3_wino43_bn_act_nhwc(x, u, sc, sh, r, relu=True), args.reps)
                outs[mode] = ops.conv3x3_wino43_bn_act_nhwc(x, u, sc, sh, r, relu=True)
                cells.append(f"{'1-D' if mode == 0 else 'nested'}:{med*1e3:7.1f}us/{fl/med/1e9:6.1f}TF")
            ops.set_tuning(ops.TUNE_WINO2D, 1)
            diff = float((outs[2] - outs[0]).abs().max() / outs[0].abs().max())
            print(f"conv {name:6s} M={n*ho*wo:8d} N={cout:4d} K={9*cin:5d}  " + "  ".join(cells) + f"  |nested - 1-D| = {diff:.2e}", flush=True)
            continue
        elif args.wino and k == 3 and s == 1:
            u = ops.wino43_transform_weights(wt)
            name = name + "w"
            med, best = timeit(lambda: ops.conv3x3_wino43_bn_act_nhwc(x, u, sc, sh, r, relu=True), args.reps)
        else:
            med, best = timeit(lambda: ops.conv2d_bn_act_nhwc(x, wt, sc, sh, r, stride=s, pad=p, relu=True), args.reps)
        fl = 2.0 * n * ho * wo * cout * k * k * (3 if name == "stem" else cin)      # algorithmic (direct) FLOP
        print(f"conv {name:7s} M={n*ho*wo:8d} N={cout:4d} K={k*k*cin:5d}  {med*1e3:8.1f} us  {fl/med/1e9:7.1f} TF (best {fl/best/1e9:6.1f})", flush=True)
    for name, m, widths, n_out in LINEAR:
        if args.only and args.only not in name:
            continue
        srcs = []
        for wd in widths:
            a = torch.randn(256 if len(widths) > 1 and m > 256 else m, wd, device=dev)
            idx = torch.randint(0, a.shape[0], (m,), device=dev) if a.shape[0] != m else None
            srcs.append((a, idx))
        kk = sum(widths)
        wt = torch.randn(n_out, kk, device=dev) * kk ** -0.5
        b = torch.randn(n_out, device=dev)
        med, best = timeit(lambda: ops.linear_gather(srcs, wt, b, m, relu=True), args.reps)
        fl = 2.0 * m * n_out * kk
        print(f"lin  {name:6s} M={m:8d} N={n_out:4d} K={kk:5d}  {med*1e3:8.1f} us  {fl/med/1e9:7.1f} TF (best {fl/best/1e9:6.1f})", flush=True)


if __name__ == "__main__":
    main()
