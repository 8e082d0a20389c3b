"""BASELINE.json configs[2]: bf16 activations + bf16 MFMA convolutions in the encoder (GNN stays fp32).

Tolerances (stated here, looser than the fp32 path's 1e-4 by design -- bf16 has an 8-bit mantissa):
  * one convolution vs F.conv2d evaluated on the SAME bf16-rounded inputs / weights in fp32: <= 1e-2 max-norm relative
    (the kernel accumulates in fp32; the difference is the final bf16 rounding of the output, 2^-9 relative);
  * whole forward (36 bf16 layers deep) vs the fp32 CPU oracle: <= 5e-2 max-norm relative on the poses.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _rand(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


@pytest.mark.parametrize("n,h,w,cin,cout,k,stride,pad,res,relu,f32out", [
    (2, 9, 11, 8, 16, 3, 1, 1, False, True, False),       # ragged M, K=72 (K tail), tiny channels
    (1, 16, 16, 8, 64, 7, 2, 3, False, True, False),      # stem shape with Cin padded to 8
    (2, 14, 14, 64, 128, 3, 2, 1, True, True, False),     # strided, residual
    (2, 14, 14, 64, 128, 1, 2, 0, False, False, False),   # downsample 1x1
    (40, 56, 56, 64, 64, 3, 1, 1, True, True, False),     # layer1 shape -> 128x64 tile
    (64, 28, 28, 128, 128, 3, 1, 1, True, True, False),   # 128x128 tile
    (37, 1, 1, 512, 2048, 1, 1, 0, False, False, True),   # the fc as a 1x1 conv on a 1x1 image, fp32 output
    (3, 13, 17, 192, 72, 3, 2, 1, True, True, False),     # ragged M and N, 3 K steps per tap (odd step count), padding taps
])
@pytest.mark.parametrize("fast", [1, 0], ids=["interleaved", "general"])
def test_conv_bf16(dev, n, h, w, cin, cout, k, stride, pad, res, relu, f32out, fast):
    """Both bf16 convolution kernels (RPG_TUNE_BF16_FAST: the interleaved buffer-load kernel serves Cin % 64 == 0, the
    general one everything else, so fast=1 exercises the dispatch and fast=0 forces the general kernel)."""
    from relpose_gnn_amd import ops
    ops.set_tuning(ops.TUNE_BF16_FAST, fast)
    x = _rand(n, cin, h, w, seed=1).bfloat16()
    wt = _rand(cout, cin, k, k, seed=2, scale=(2.0 / (cin * k * k)) ** 0.5).bfloat16()
    scale = torch.rand(cout, generator=torch.Generator().manual_seed(3)) + 0.5
    shift = _rand(cout, seed=4, scale=0.1)
    ref = F.conv2d(x.float(), wt.float(), None, stride=stride, padding=pad) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    r = None
    if res:
        r = _rand(*ref.shape, seed=5).bfloat16()
        ref = ref + r.float()
    if relu:
        ref = F.relu(ref)
    y = ops.conv2d_bn_act_nhwc_bf16(x.permute(0, 2, 3, 1).contiguous().to(dev), wt.permute(0, 2, 3, 1).contiguous().to(dev),
                                    scale.to(dev), shift.to(dev), None if r is None else r.permute(0, 2, 3, 1).contiguous().to(dev),
                                    stride=stride, pad=pad, relu=relu, out_f32=f32out)
    ops.set_tuning(ops.TUNE_BF16_FAST, 1)
    assert y.dtype == (torch.float32 if f32out else torch.bfloat16)
    err = rel_err(y.float().cpu().permute(0, 3, 1, 2), ref)
    assert err < (1e-5 if f32out else 1e-2), err


@pytest.mark.parametrize("tile", [0, 1, 2, 3], ids=["64x64", "128x128", "256x64", "128x64"])
def test_conv_bf16_interleaved_tiles(dev, tile):
    """Every tile of the interleaved bf16 kernel (RPG_TUNE_BF16_TILE) on a shape with a residual, ragged rows (M % 256 != 0),
    a ragged channel tile (Cout = 72) and an odd number of K steps per workgroup."""
    from relpose_gnn_amd import ops
    n, h, w, cin, cout = 5, 23, 19, 64, 72
    x = _rand(n, cin, h, w, seed=21).bfloat16()
    wt = _rand(cout, cin, 3, 3, seed=22, scale=(2.0 / (cin * 9)) ** 0.5).bfloat16()
    scale = torch.rand(cout, generator=torch.Generator().manual_seed(23)) + 0.5
    shift = _rand(cout, seed=24, scale=0.1)
    r = _rand(n, cout, h, w, seed=25).bfloat16()
    ref = F.relu(F.conv2d(x.float(), wt.float(), None, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + r.float())
    ops.set_tuning(ops.TUNE_BF16_TILE, tile)
    try:
        y = ops.conv2d_bn_act_nhwc_bf16(x.permute(0, 2, 3, 1).contiguous().to(dev), wt.permute(0, 2, 3, 1).contiguous().to(dev),
                                        scale.to(dev), shift.to(dev), r.permute(0, 2, 3, 1).contiguous().to(dev), stride=1, pad=1,
                                        relu=True)
    finally:
        ops.set_tuning(ops.TUNE_BF16_TILE, -1)
    err = rel_err(y.float().cpu().permute(0, 3, 1, 2), ref)
    assert err < 1e-2, err


def test_bf16_encoder_forward_vs_fp32_oracle(dev):
    """configs[2] shape family at test size: R3 dims, 224x224, 2 graphs x 8 nodes; bf16 encoder + fp32 GNN."""
    import json
    import relpose_gnn_amd.synth as S
    from oracle import posenet_ref as O
    from relpose_gnn_amd.graph import fc_batch
    from relpose_gnn_amd.posenet import PoseNetX_R2
    from relpose_gnn_amd.resnet import resnet34
    D = 2048
    m = PoseNetX_R2(resnet34(), droprate=0.0, pretrained=False, feat_dim=D, edge_feat_dim=D, node_dim=D,
                    input_img_height=224, use_gnn=True, knn=-1, use_AP=True, gnn_recursion=2)
    sd = S.synth_state_dict(S.posenet_r2_param_shapes(D, D, D), seed=1)
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    x = S.synth_images(16, 224, 224, seed=6)
    d = fc_batch(x, 8).to(dev)
    a32, r32, _ = m(d)
    m.encoder_dtype = "bf16"
    assert m.encoder_dtype == "bf16"
    a, r, _ = m(d)
    feat = m._enc.run(m.feature_extractor.state_dict, "", x.view(16, 3, 224, 224).to(dev))
    st = {}
    oa, orr, _ = O.posenet_forward(sd, x, d.edge_index.cpu(), 224, 2, st)
    ef, ea, er = rel_err(feat.cpu(), st["fc"]), rel_err(a.cpu(), oa), rel_err(r.cpu(), orr)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "parity_report.jsonl"), "a") as f:
            f.write(json.dumps({"case": "bf16_encoder_R3_224px_2x8node_vs_fp32_oracle", "feat_rel_err": ef,
                                "abs_pose_rel_err": ea, "rel_pose_rel_err": er}) + "\n")
    assert ef < 5e-2 and ea < 5e-2 and er < 5e-2, (ef, ea, er)
    m.encoder_dtype = "f32"                           # and back: the fp32 path is unaffected
    a2, r2, _ = m(d)
    assert torch.equal(a2, a32) and torch.equal(r2, r32)


def test_bf16_forward_with_persistent_layer1_is_bit_identical(dev):
    """RPG_TUNE_BF16_PERSIST = 1 inside the model: 6 graphs x 8 nodes x 224x224 put 294 tiles of 512 pixels on layer 1 (more than one
    round of the 256 CUs: the persistent form is dispatched); the poses equal the default dispatch bit for bit."""
    import relpose_gnn_amd.synth as S
    from relpose_gnn_amd import ops
    from relpose_gnn_amd.graph import fc_batch
    from relpose_gnn_amd.posenet import PoseNetX_R2
    from relpose_gnn_amd.resnet import resnet34
    D = 2048
    m = PoseNetX_R2(resnet34(), droprate=0.0, pretrained=False, feat_dim=D, edge_feat_dim=D, node_dim=D,
                    input_img_height=224, use_gnn=True, knn=-1, use_AP=True, gnn_recursion=2)
    m.load_state_dict(S.synth_state_dict(S.posenet_r2_param_shapes(D, D, D), seed=1))
    m = m.to(dev).eval()
    m.encoder_dtype = "bf16"
    d = fc_batch(S.synth_images(48, 224, 224, seed=9), 8).to(dev)
    a0, r0, _ = m(d)
    ops.set_tuning(ops.TUNE_BF16_PERSIST, 1)
    try:
        a1, r1, _ = m(d)
    finally:
        ops.set_tuning(ops.TUNE_BF16_PERSIST, 0)
    assert torch.equal(a0, a1) and torch.equal(r0, r1)
    assert bool(torch.isfinite(a1).all()) and float(a1.abs().max()) > 0


@pytest.mark.parametrize("m,k,n_out,gather", [(333, 192, 100, 2), (1792, 2048, 768, 0), (130, 256, 2048, 1), (64, 72, 40, 0)])
def test_linear_bf16(dev, m, k, n_out, gather):
    """rpg_f32_to_bf16 + rpg_linear_bf16 (bf16 inputs / weights, fp32 accumulate / bias / residual / output; plain and
    gathered fp32 residual rows; K a multiple of 64 -> interleaved kernel, else the general one) vs torch on the
    bf16-rounded operands (so the tolerance only covers accumulation order)."""
    from relpose_gnn_amd import ops
    a = _rand(m, k, seed=1)
    w = _rand(n_out, k, seed=2, scale=k ** -0.5)
    bias = _rand(n_out, seed=3)
    ab = ops.f32_to_bf16(a.to(dev))
    assert torch.equal(ab.cpu(), a.bfloat16())
    wide = torch.zeros(m, 2 * k, dtype=torch.bfloat16, device=dev)            # column-offset form used for concatenations
    ops.f32_to_bf16(a.to(dev), out=wide, col_off=k)
    assert torch.equal(wide[:, k:].cpu(), a.bfloat16()) and float(wide[:, :k].float().abs().max()) == 0.0
    ref = F.linear(a.bfloat16().float(), w.bfloat16().float(), bias)
    res = idx = res2 = idx2 = None
    if gather == 0:
        res = _rand(m, n_out, seed=4)
        ref = ref + res
    else:
        g = torch.Generator().manual_seed(5)
        table = _rand(50, 3 * n_out, seed=6)                                   # rows of [r1 | r2 | unused], pitch 3 * n_out
        idx = torch.randint(0, 50, (m,), generator=g)
        res = table
        ref = ref + table[idx, :n_out]
        if gather == 2:
            idx2 = torch.randint(0, 50, (m,), generator=g)
            res2 = table[:, n_out:]                                            # a view at column offset n_out
            ref = ref + table[idx2, n_out:2 * n_out]
    ref = F.relu(ref)
    tdev = None if res is None else res.to(dev)
    r2dev = None
    if res2 is not None:
        r2dev = tdev[:, n_out:]                                                # same storage, column offset (non-contiguous view)
    out = _linear_bf16_raw(ops, ab, w.bfloat16().to(dev), bias.to(dev), tdev, None if idx is None else idx.to(dev), r2dev,
                           None if idx2 is None else idx2.to(dev), 3 * n_out if gather else n_out, m, k, n_out)
    assert rel_err(out.cpu(), ref) < 2e-5


@pytest.mark.parametrize("cfg", [0, 1, 3, 7, 8])
@pytest.mark.parametrize("m,k,n_out,gather", [(3584, 2048, 2048, 2), (1792, 4096, 2048, 0), (1801, 2048, 2056, 1)])
def test_linear_bf16_on_dma_configs(dev, cfg, m, k, n_out, gather):
    """RPG_TUNE_BF16_LINEAR_DMA = 10 + cfg: the edge-row Linears of the bf16 GNN on a configuration of the LDS-DMA kernel (a
    1 x 1 convolution over an m-pixel image) -- plain / gathered / twice-gathered fp32 residual rows through the general
    epilogue, ragged m and n_out; same reference and bar as test_linear_bf16."""
    from relpose_gnn_amd import ops
    a = _rand(m, k, seed=1)
    w = _rand(n_out, k, seed=2, scale=k ** -0.5)
    bias = _rand(n_out, seed=3)
    ab = ops.f32_to_bf16(a.to(dev))
    ref = F.linear(a.bfloat16().float(), w.bfloat16().float(), bias)
    idx = res2 = idx2 = None
    if gather == 0:
        res = _rand(m, n_out, seed=4)
        ref = ref + res
    else:
        g = torch.Generator().manual_seed(5)
        res = _rand(50, 3 * n_out, seed=6)
        idx = torch.randint(0, 50, (m,), generator=g)
        ref = ref + res[idx, :n_out]
        if gather == 2:
            idx2 = torch.randint(0, 50, (m,), generator=g)
            ref = ref + res[idx2, n_out:2 * n_out]
    ref = F.relu(ref)
    tdev = res.to(dev)
    r2dev = tdev[:, n_out:] if gather == 2 else None
    args = (ops, ab, w.bfloat16().to(dev), bias.to(dev), tdev, None if idx is None else idx.to(dev), r2dev,
            None if idx2 is None else idx2.to(dev), 3 * n_out if gather else n_out, m, k, n_out)
    base = _linear_bf16_raw(*args)
    ops.set_tuning(ops.TUNE_BF16_LINEAR_DMA, 10 + cfg)
    try:
        out = _linear_bf16_raw(*args)
    finally:
        ops.set_tuning(ops.TUNE_BF16_LINEAR_DMA, 0)
    assert rel_err(out.cpu(), ref) < 2e-5 and rel_err(base.cpu(), ref) < 2e-5


@pytest.mark.parametrize("cfg", range(10))
@pytest.mark.parametrize("n,h,w,cin,cout,k,stride,pad,res,relu,f32out", [
    (40, 56, 56, 64, 64, 3, 1, 1, True, True, False),     # layer-1 shape: 1 tap = 1 K step of 64, many M tiles, N = 64 < BN
    (24, 28, 28, 128, 128, 3, 1, 1, True, True, False),   # layer-2 shape
    (9, 14, 14, 256, 256, 3, 1, 1, False, True, False),   # layer-3 shape: ragged M (1764 rows), 36 K steps of 64
    (3, 13, 17, 192, 72, 3, 2, 1, True, True, False),     # ragged M and N, strided, 3 K steps per tap (odd), padding taps
    (2, 14, 14, 64, 128, 1, 2, 0, False, False, False),   # downsample 1x1: a single K step (fewer steps than LDS images)
    (37, 1, 1, 512, 2048, 1, 1, 0, False, False, True),   # the fc as a 1x1 conv on a 1x1 image, fp32 output
    (5, 9, 11, 96, 40, 3, 1, 1, True, False, False),      # Cin = 96: only the 32-wide K step configurations apply
])
def test_conv_bf16_dma_configs(dev, cfg, n, h, w, cin, cout, k, stride, pad, res, relu, f32out):
    """Every configuration of the LDS-DMA bf16 convolution kernel (RPG_TUNE_BF16_DMA = 10 + cfg: tile, wave grid, K step, number
    of LDS images) on shapes that exercise its edges: zero-filled out-of-image taps and ragged rows (hardware zero fill of the
    DMA), N smaller than the tile, fewer K steps than pipeline stages, stride 2, the swizzled source chunks.  Against F.conv2d
    on the same bf16 inputs in fp32; where a configuration is not eligible (Cin % K step) the launcher falls back and the check
    still holds."""
    from relpose_gnn_amd import ops
    x = _rand(n, cin, h, w, seed=1).bfloat16()
    wt = _rand(cout, cin, k, k, seed=2, scale=(2.0 / (cin * k * k)) ** 0.5).bfloat16()
    scale = torch.rand(cout, generator=torch.Generator().manual_seed(3)) + 0.5
    shift = _rand(cout, seed=4, scale=0.1)
    ref = F.conv2d(x.float(), wt.float(), None, stride=stride, padding=pad) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    r = None
    if res:
        r = _rand(*ref.shape, seed=5).bfloat16()
        ref = ref + r.float()
    if relu:
        ref = F.relu(ref)
    ops.set_tuning(ops.TUNE_BF16_DMA, 10 + cfg)
    try:
        y = ops.conv2d_bn_act_nhwc_bf16(x.permute(0, 2, 3, 1).contiguous().to(dev), wt.permute(0, 2, 3, 1).contiguous().to(dev),
                                        scale.to(dev), shift.to(dev), None if r is None else r.permute(0, 2, 3, 1).contiguous().to(dev),
                                        stride=stride, pad=pad, relu=relu, out_f32=f32out)
    finally:
        ops.set_tuning(ops.TUNE_BF16_DMA, 1)
    assert rel_err(y.float().cpu().permute(0, 3, 1, 2), ref) < (1e-4 if f32out else 1e-2)


@pytest.mark.parametrize("mode", [2, 3, 12], ids=["by_width", "tile256x128", "by_width_3stages"])
@pytest.mark.parametrize("n,h,w,cin,cout,res,relu", [
    (40, 56, 56, 64, 64, True, True),       # layer-1 shape: 10 patch rows of 64 slots, 2 chunks
    (24, 28, 28, 128, 128, True, True),     # layer-2 shape
    (9, 14, 14, 256, 256, False, True),     # layer-3 shape: tiles span 2 images, ragged M (1764)
    (70, 7, 7, 512, 512, True, True),       # layer-4 shape: tiles span 6-7 images (zero rows between them), 16 chunks
    (3, 13, 17, 192, 72, True, False),      # odd sizes, ragged N, 6 chunks
    (5, 8, 11, 64, 320, False, True),       # 256x341's last stage; N = 320: two channel tiles, the second ragged
    (2, 5, 3, 128, 128, True, True),        # tiny image: 3 of 16 slots per patch row used, the whole batch inside one tile
    (1, 64, 86, 64, 64, False, True),       # 256x341's first stage: 96 slots per row
    (256, 14, 14, 256, 256, True, True),    # 196 tiles of 256 x 256: the wide tile (and its four-stage form) is dispatched
    (300, 7, 7, 512, 512, False, True),     # 58 x 2 tiles of 256 x 256 is too few -> 256 x 128; 53-KB patches: three stages only
])
def test_conv3x3_bf16_patch_kernel(dev, mode, n, h, w, cin, cout, res, relu):
    _patch_kernel_case(dev, mode, 0, n, h, w, cin, cout, res, relu)


@pytest.mark.parametrize("n,h,w,cin,cout,res,relu", [
    (70, 56, 56, 64, 64, True, True),       # layer 1: 429 tiles of 512 rows on 256 CUs (1.7 rounds: workgroups with 1 and with 2 tiles), ragged last tile
    (171, 56, 56, 64, 64, False, True),     # 1048 tiles: 4.1 rounds, the stage rotation of the four-stage form wraps across tiles
    (100, 40, 40, 64, 64, True, True),      # 48 slots per row, 18 patch rows: 313 tiles, ragged
    (300, 13, 17, 64, 64, True, False),     # small odd images: tiles span 3 images, 130 tiles only -> not eligible, one workgroup per tile
    (200, 28, 28, 128, 40, False, True),    # 307 tiles, 4 chunks per tile (the wrap happens in front of chunk 3), ragged N = 40 inside the one channel tile
])
def test_conv3x3_bf16_patch_kernel_persistent(dev, n, h, w, cin, cout, res, relu):
    """RPG_TUNE_BF16_PERSIST = 1: the persistent form of the patch kernel on <= 64 output channels (one workgroup per CU
    walks its tiles, the next tile's first patch chunk and weights load during the current tile's last chunk and epilogue, slabs
    beside patch buffer 1) against F.conv2d, and bit-identical to the one-workgroup-per-tile form (same MFMA order per tile)."""
    _patch_kernel_case(dev, 2, 1, n, h, w, cin, cout, res, relu)


def _patch_kernel_case(dev, mode, persist, n, h, w, cin, cout, res, relu):
    """The patch kernel of the bf16 encoder (3x3 / stride 1 / pad 1 with the input patch resident in LDS; RPG_TUNE_BF16_PATCH = 2:
    every eligible size, tile by output width -- 512 x 64, 512 x 128, 256 x 128, 256 x 256; 3: the 256 x 128 tile everywhere; 12:
    as 2 but never the four-weight-stage form) against F.conv2d on the same bf16 inputs in fp32:
    tiles that span image rows and whole images (virtual zero rows), zero halo slots from out-of-range DMA lanes, ragged M / N,
    2 .. 16 channel chunks through the two patch buffers and three weight stages."""
    from relpose_gnn_amd import ops
    x = _rand(n, cin, h, w, seed=31).bfloat16()
    wt = _rand(cout, cin, 3, 3, seed=32, scale=(2.0 / (cin * 9)) ** 0.5).bfloat16()
    scale = torch.rand(cout, generator=torch.Generator().manual_seed(33)) + 0.5
    shift = _rand(cout, seed=34, scale=0.1)
    ref = F.conv2d(x.float(), wt.float(), None, stride=1, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    r = None
    if res:
        r = _rand(*ref.shape, seed=35).bfloat16()
        ref = ref + r.float()
    if relu:
        ref = F.relu(ref)
    ops.set_tuning(ops.TUNE_BF16_PATCH, mode)
    args = (x.permute(0, 2, 3, 1).contiguous().to(dev), wt.permute(0, 2, 3, 1).contiguous().to(dev), scale.to(dev), shift.to(dev),
            None if r is None else r.permute(0, 2, 3, 1).contiguous().to(dev))
    try:
        ops.set_tuning(ops.TUNE_BF16_PERSIST, 0)
        y = ops.conv2d_bn_act_nhwc_bf16(*args, stride=1, pad=1, relu=relu)
        if persist:
            ops.set_tuning(ops.TUNE_BF16_PERSIST, 1)
            y1 = ops.conv2d_bn_act_nhwc_bf16(*args, stride=1, pad=1, relu=relu)
            assert torch.equal(y1, y)
            y = y1
    finally:
        ops.set_tuning(ops.TUNE_BF16_PATCH, 1)
        ops.set_tuning(ops.TUNE_BF16_PERSIST, 0)
    got = y.float().cpu().permute(0, 3, 1, 2)
    assert rel_err(got, ref) < 1e-2
    assert float((got - ref).abs().mean() / ref.abs().mean().clamp(min=1e-30)) < 3e-3     # bf16 output rounding only


@pytest.mark.parametrize("n,h,w,cin,cout,k,stride,pad,res", [
    (100, 56, 56, 64, 64, 3, 1, 1, True),       # layer 1 at 100 images: LDS-DMA kernel 256 x 64 (two workgroups per CU), ragged last tile
    (24, 28, 28, 128, 128, 3, 1, 1, True),      # layer 2: patch kernel 512 x 128
    (40, 14, 14, 256, 256, 3, 1, 1, False),     # layer 3: patch kernel 256 x 256, tiles span images
    (70, 7, 7, 512, 512, 3, 1, 1, True),        # layer 4
    (24, 56, 56, 64, 128, 3, 2, 1, False),      # 3x3 / stride 2
    (24, 56, 56, 64, 128, 1, 2, 0, False),      # 1x1 / stride 2 downsample (no ReLU in the model; here with)
    (3, 13, 17, 192, 72, 3, 1, 1, True),        # ragged N (72 = 2 x 32 + 8), odd sizes
    (5, 8, 11, 64, 320, 3, 1, 1, True),         # two channel tiles, the second ragged
])
def test_conv_bf16_lean_epilogue_equals_general(dev, n, h, w, cin, cout, k, stride, pad, res):
    """The branch-free epilogue of the plain bf16 convolutions (round 4, RPG_TUNE_BF16_LEAN_EPI = 1: raw buffer accesses with
    out-of-range offsets for invalid rows / columns, 32-bit offsets, residual as a template parameter) against the general
    epilogue of rounds 1-3 (= 0) on the same operands: the same arithmetic in the same order, so the bf16 outputs agree
    bit for bit; and both within the bf16 bar of F.conv2d.  Every kernel family that reaches it: LDS-DMA, patch, ragged tiles."""
    from relpose_gnn_amd import ops
    x = _rand(n, cin, h, w, seed=61).bfloat16()
    wt = _rand(cout, cin, k, k, seed=62, scale=(2.0 / (cin * k * k)) ** 0.5).bfloat16()
    scale = torch.rand(cout, generator=torch.Generator().manual_seed(63)) + 0.5
    shift = _rand(cout, seed=64, scale=0.1)
    ref = F.conv2d(x.float(), wt.float(), None, stride=stride, padding=pad) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    r = None
    if res:
        r = _rand(*ref.shape, seed=65).bfloat16()
        ref = ref + r.float()
    ref = F.relu(ref)
    xd, wd = x.permute(0, 2, 3, 1).contiguous().to(dev), wt.permute(0, 2, 3, 1).contiguous().to(dev)
    rd = None if r is None else r.permute(0, 2, 3, 1).contiguous().to(dev)
    outs = {}
    try:
        for lean in (1, 0):
            ops.set_tuning(ops.TUNE_BF16_LEAN_EPI, lean)
            outs[lean] = ops.conv2d_bn_act_nhwc_bf16(xd, wd, scale.to(dev), shift.to(dev), rd, stride=stride, pad=pad, relu=True)
    finally:
        ops.set_tuning(ops.TUNE_BF16_LEAN_EPI, 1)
    assert torch.equal(outs[1], outs[0])
    got = outs[1].float().cpu().permute(0, 3, 1, 2)
    assert rel_err(got, ref) < 1e-2
    assert float((got - ref).abs().mean() / ref.abs().mean().clamp(min=1e-30)) < 3e-3


@pytest.mark.parametrize("kernel", [1, 33, 1 + (7 << 8), 33 + (14 << 8), 3])       # RPG_TUNE_FUSED_STEM: strip-march kernel (default: both halves per wave / one half per wave / 7- and 14-row bands), tile kernel
@pytest.mark.parametrize("n,h,w", [(2, 224, 224), (1, 256, 341), (3, 37, 53), (2, 9, 5), (1, 64, 500), (5, 32, 40), (1, 1, 1), (70, 64, 72),
                                   (67, 40, 24), (2, 250, 123), (1, 30, 130)])
def test_fused_stem_bf16(dev, n, h, w, kernel):
    """rpg_stem_conv7x7s2_bn_relu_maxpool_bf16 (fp32 NCHW in -> pooled bf16 NHWC out, bf16 MFMA) vs conv2d(7x7, s2, p3) on the
    SAME bf16-rounded input and weights in fp32 -> BN affine -> ReLU -> max_pool2d(3, 2, 1) -> bf16 (the torchvision stem
    reached from posenet.py:1037).  What is left between the two is fp32 summation order and, where that moves a value across a
    bf16 rounding boundary, one bf16 ulp (2^-8 relative) on single elements: bar 1e-2 max-norm like every bf16 convolution here,
    and the MEAN error must sit at the fp32-noise level (1e-4): a wrong tap or window would move every output.
    224x224 (two column tiles of 28), 256x341 (four ragged column tiles), odd / tiny sizes, a wide image (five column tiles), and
    >= 64 images not a multiple of 8 (the per-XCD image order of the kernel, ragged).
    Round 6: the strip-march kernel (default; csrc/stem_bf16.hip stem_strip_bf16_kernel -- strips of 15 pooled columns, bands of 14
    pooled rows: 224 -> 4 x 4, 341 wide -> 6 ragged strips, odd convolution heights (250 -> 125 rows: the last pooled row ends on
    an even convolution row), images narrower than a strip) and the tile kernel of rounds 3-5 behind RPG_TUNE_FUSED_STEM = 3."""
    from relpose_gnn_amd import ops
    from relpose_gnn_amd.params import pack_stem_bf16
    ops.set_tuning(ops.TUNE_FUSED_STEM, kernel)
    try:
        _fused_stem_bf16_case(dev, n, h, w)
    finally:
        ops.set_tuning(ops.TUNE_FUSED_STEM, 1)


def _fused_stem_bf16_case(dev, n, h, w):
    from relpose_gnn_amd import ops
    from relpose_gnn_amd.params import pack_stem_bf16
    x = _rand(n, 3, h, w, seed=h)
    wt = _rand(64, 3, 7, 7, seed=2, scale=(2.0 / 147) ** 0.5)
    g = torch.Generator().manual_seed(3)
    scale = (torch.rand(64, generator=g) + 0.5) * torch.where(torch.rand(64, generator=g) < 0.15, -1.0, 1.0)   # some negative gammas
    shift = _rand(64, seed=4, scale=0.3)
    conv = F.conv2d(x.bfloat16().float(), wt.bfloat16().float(), None, stride=2, padding=3)
    ref = F.max_pool2d(F.relu(conv * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)), 3, 2, 1).bfloat16().float()
    y = ops.stem_conv_bn_relu_maxpool_bf16(x.to(dev), pack_stem_bf16(wt).to(dev), scale.to(dev), shift.to(dev))
    assert y.dtype == torch.bfloat16 and y.shape == (n, ref.shape[2], ref.shape[3], 64)
    got = y.float().cpu().permute(0, 3, 1, 2)
    assert rel_err(got, ref) < 1e-2
    assert float((got - ref).abs().mean() / ref.abs().mean().clamp(min=1e-30)) < 2e-4
    # images already rounded to bf16 (host-side staging of evaluate_stream): the same kernel on a bf16 input, bit-identical
    y16 = ops.stem_conv_bn_relu_maxpool_bf16(x.bfloat16().to(dev), pack_stem_bf16(wt).to(dev), scale.to(dev), shift.to(dev))
    assert torch.equal(y16, y)
    # and the unfused three-kernel chain of the same encoder (re-layout + generic conv on 8 channels + max-pool) agrees
    w8 = F.pad(wt.permute(0, 2, 3, 1), (0, 5)).bfloat16().contiguous().to(dev)
    x8 = F.pad(x.permute(0, 2, 3, 1), (0, 5)).bfloat16().contiguous().to(dev)
    c3 = ops.conv2d_bn_act_nhwc_bf16(x8, w8, scale.to(dev), shift.to(dev), None, stride=2, pad=3, relu=True)
    ref3 = F.max_pool2d(c3.float().cpu().permute(0, 3, 1, 2), 3, 2, 1)
    assert rel_err(got, ref3) < 1e-2


def _linear_bf16_raw(ops, a, w, bias, res, idx, res2, idx2, ldr, m, k, n_out):
    """Calls the C entry point directly (residual2 is a column-offset view into the same table, which the tensor-level
    wrapper would copy)."""
    from relpose_gnn_amd import _lib as L
    out = torch.empty((m, n_out), dtype=torch.float32, device=a.device)
    p = lambda t: None if t is None else t.data_ptr()
    L.check(L.lib().rpg_linear_bf16(p(a), p(w), p(bias), p(res), p(idx), p(res2), p(idx2), ldr, p(out), m, k, n_out, 1,
                                    torch.cuda.current_stream().cuda_stream), "linear_bf16")
    return out


def test_bf16_gnn_forward_vs_fp32(dev):
    """gnn_dtype = 'bf16' (Linears of the GNN on the bf16 matrix pipe) on top of the bf16 encoder, R3 dims, 224x224,
    2 graphs x 8 nodes: against the fp32 oracle, same 5e-2 bar as the bf16 encoder alone, and the GNN-only effect against
    the fp32 GNN on identical features."""
    import json
    import relpose_gnn_amd.synth as S
    from oracle import posenet_ref as O
    from relpose_gnn_amd.graph import fc_batch
    from relpose_gnn_amd.posenet import PoseNetX_R2
    from relpose_gnn_amd.resnet import resnet34
    D = 2048
    m = PoseNetX_R2(resnet34(), droprate=0.0, pretrained=False, feat_dim=D, edge_feat_dim=D, node_dim=D,
                    input_img_height=224, use_gnn=True, knn=-1, use_AP=True, gnn_recursion=2)
    sd = S.synth_state_dict(S.posenet_r2_param_shapes(D, D, D), seed=1)
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    x = S.synth_images(16, 224, 224, seed=6)
    d = fc_batch(x, 8).to(dev)
    a32, r32, _ = m(d)                                                         # fp32 encoder + fp32 GNN
    m.gnn_dtype = "bf16"
    ag, rg, _ = m(d)                                                           # fp32 encoder + bf16 GNN
    e_gnn = (rel_err(ag.cpu(), a32.cpu()), rel_err(rg.cpu(), r32.cpu()))
    m.encoder_dtype = "bf16"
    a, r, _ = m(d)                                                             # bf16 encoder + bf16 GNN
    oa, orr, _ = O.posenet_forward(sd, x, d.edge_index.cpu(), 224, 2, {})
    ea, er = rel_err(a.cpu(), oa), rel_err(r.cpu(), orr)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "parity_report.jsonl"), "a") as f:
            f.write(json.dumps({"case": "bf16_encoder_and_bf16_gnn_R3_224px_2x8node_vs_fp32_oracle", "abs_pose_rel_err": ea,
                                "rel_pose_rel_err": er, "gnn_only_abs": e_gnn[0], "gnn_only_rel": e_gnn[1]}) + "\n")
    # measured: GNN alone 4.3e-2 / 2.2e-2 (abs / rel poses; ~20 chained bf16-input GEMMs), with the bf16 encoder vs the
    # fp32 oracle 3.8e-2 / 2.1e-2
    assert max(e_gnn) < 6e-2, e_gnn
    assert ea < 5e-2 and er < 5e-2, (ea, er)
    m.gnn_dtype, m.encoder_dtype = "f32", "f32"
    a2, r2, _ = m(d)
    assert torch.equal(a2, a32) and torch.equal(r2, r32)


@pytest.mark.parametrize("shape", [(3, 56, 56), (7, 14, 56), (5, 24, 40), (1, 56, 56), (40, 56, 56),
                                   (2, 64, 86), (3, 43, 64), (9, 64, 86), (2, 20, 85), (1, 30, 120), (3, 30, 63)])
def test_fused_basicblock64_equals_two_convolutions(dev, shape):
    """Round 5: conv1 + BN + ReLU + conv2 + BN + identity + ReLU of a 64-channel BasicBlock as ONE kernel
    with the intermediate in LDS (csrc/block_bf16.inc; torchvision BasicBlock reached from modules/posenet.py:1037).  The
    intermediate is rounded to bf16 exactly where the two-launch path stores it, so the outputs must be BIT-IDENTICAL to two
    rpg_conv2d_bn_act_nhwc_bf16 calls: tiles straddling image boundaries (14-row images), a 40-wide map (48-slot patch rows),
    ragged last tiles, a single image, and 40 images (245 tiles).
    Round 6: maps wider than 63 pixels run as two column strips per image (virtual images of ceil(W / 2) + 2
    columns, the two columns next to the cut recomputed and not stored): the 64 x 86 layer-1 maps of the 256 x 341 evaluation
    shape (datasets/dataset_7Scenes_multi.py:341,434), 43 x 64, odd widths (85: the strips' shares differ by one column), the
    widest map taken (120), and 63 (one column too wide for a single strip's 64-slot patch rows: two strips)."""
    from relpose_gnn_amd import ops
    n, h, w = shape
    g = torch.Generator().manual_seed(1000 + n * h + w)
    x = torch.randn((n, h, w, 64), generator=g).bfloat16().to(dev)
    w1 = (torch.randn((64, 3, 3, 64), generator=g) * (2.0 / 576) ** 0.5).bfloat16().to(dev)
    w2 = (torch.randn((64, 3, 3, 64), generator=g) * (2.0 / 576) ** 0.5).bfloat16().to(dev)
    s1, b1 = (torch.rand(64, generator=g) + 0.5).to(dev), (torch.randn(64, generator=g) * 0.2).to(dev)
    s2, b2 = (torch.rand(64, generator=g) + 0.5).to(dev), (torch.randn(64, generator=g) * 0.2).to(dev)
    # the two-launch reference on the PATCH kernel (RPG_TUNE_BF16_PATCH = 2: wherever eligible; the dispatcher picks it by itself
    # from 21 images of 56x56 up): the fused kernel walks K in the patch kernel's order (32-channel chunk major, tap minor);
    # below that size the dispatcher's im2col kernel sums tap major and differs in the last bf16 bit of ~0.04 % of the outputs
    def two_launches(xx):
        t = ops.conv2d_bn_act_nhwc_bf16(xx, w1, s1, b1, None, stride=1, pad=1, relu=True)
        return ops.conv2d_bn_act_nhwc_bf16(t, w2, s2, b2, xx, stride=1, pad=1, relu=True)

    ops.set_tuning(ops.TUNE_BF16_PATCH, 2)
    try:
        if w <= 62:
            want = two_launches(x)
        else:
            # Two-strip launches: the bit-exact reference is the patch-kernel pair on each strip's VIEW of the image (the patch
            # kernel does not take 86-wide maps -- more than 64 patch pieces --, and the im2col kernel the dispatcher uses there sums
            # tap major: last-bit differences in ~0.04 % of the outputs).  View = ceil(w / 2) + 2 columns ending at the image edge;
            # the strip's own share of the columns is compared, the two columns next to the cut are recomputed by both strips.
            ws = (w + 1) // 2
            wv = ws + 2
            y0 = two_launches(x[:, :, :wv, :].contiguous())
            y1 = two_launches(x[:, :, w - wv:, :].contiguous())
            want = torch.cat([y0[:, :, :ws, :], y1[:, :, ws - (w - wv):, :]], dim=2)
            whole = two_launches(x)                       # what the encoder runs with the fusion off: equal up to summation order
    finally:
        ops.set_tuning(ops.TUNE_BF16_PATCH, 1)
    got = ops.basicblock64_bf16(x, w1, s1, b1, w2, s2, b2)
    torch.cuda.synchronize()
    assert got.shape == want.shape and bool(torch.isfinite(got.float()).all())
    bad = (got != want).nonzero()
    assert bad.numel() == 0, (shape, int(bad.shape[0]), bad[:5].tolist(), float((got.float() - want.float()).abs().max()))
    if w > 62:
        off = got != whole
        assert float(off.float().mean()) < 2e-3 and rel_err(got.float(), whole.float()) < 1e-2
    # and against an fp32 reference of the block on the same bf16 operands (bf16 bar of this file's convolution tests)
    xf = x.float().permute(0, 3, 1, 2).cpu()
    tf = torch.relu(torch.nn.functional.conv2d(xf, w1.float().permute(0, 3, 1, 2).cpu(), padding=1) * s1.cpu().view(1, -1, 1, 1) + b1.cpu().view(1, -1, 1, 1))
    tf = tf.bfloat16().float()                              # the intermediate is stored in bf16 by both paths
    yf = torch.relu(torch.nn.functional.conv2d(tf, w2.float().permute(0, 3, 1, 2).cpu(), padding=1) * s2.cpu().view(1, -1, 1, 1) + b2.cpu().view(1, -1, 1, 1) + xf)
    assert rel_err(got.float().cpu().permute(0, 3, 1, 2), yf) < 1e-2


@pytest.mark.parametrize("shape", [(300, 64, 86), (96, 56, 56)])
def test_fused_basicblock64_large_batches_and_persistent_form(dev, shape):
    """Round 6: (i) regression -- the two-strip epilogue split the pixel index m into (virtual image, row, column) with a
    multiply-high 'magic' division of m itself, exact only below 2^32 / (2880 - 2^32 mod 2880) = 1 636 801 pixels on 64 x 86 maps:
    from real image 284 on, the LAST pixel of every second virtual image came out one image too high and its store was dropped (the
    output kept whatever the allocation held).  The division is tile-relative now; 300 images of 64 x 86 must equal the same images
    run 12 at a time, bit for bit, with the output buffers poisoned by the allocator's reuse pattern being irrelevant.
    (ii) the PERSISTENT form of the kernel (RPG_TUNE_BF16_FUSE_BLOCK = 3: one workgroup per CU walks tiles b, b + 256, ..., the next
    tile's first loads under the current tile's last nine steps) is bit-identical to one tile per workgroup."""
    from relpose_gnn_amd import ops
    n, h, w = shape
    g = torch.Generator().manual_seed(77 + w)
    x = torch.randn((n, h, w, 64), generator=g).bfloat16().to(dev)
    w1 = (torch.randn((64, 3, 3, 64), generator=g) * (2.0 / 576) ** 0.5).bfloat16().to(dev)
    w2 = (torch.randn((64, 3, 3, 64), generator=g) * (2.0 / 576) ** 0.5).bfloat16().to(dev)
    s1, b1 = (torch.rand(64, generator=g) + 0.5).to(dev), (torch.randn(64, generator=g) * 0.2).to(dev)
    s2, b2 = (torch.rand(64, generator=g) + 0.5).to(dev), (torch.randn(64, generator=g) * 0.2).to(dev)
    try:
        ops.set_tuning(ops.TUNE_BF16_FUSE_BLOCK, 1)
        parts = torch.cat([ops.basicblock64_bf16(x[i:i + 12].contiguous(), w1, s1, b1, w2, s2, b2) for i in range(0, n, 12)])
        # fill the allocator's free blocks of this size with NaN patterns: a dropped store then shows
        junk = torch.full_like(x, float("nan"))
        del junk
        one = ops.basicblock64_bf16(x, w1, s1, b1, w2, s2, b2)
        ops.set_tuning(ops.TUNE_BF16_FUSE_BLOCK, 3)
        junk = torch.full_like(x, float("nan"))
        del junk
        per = ops.basicblock64_bf16(x, w1, s1, b1, w2, s2, b2)
        ops.set_tuning(ops.TUNE_BF16_FUSE_BLOCK, 5)      # (iii) the two-group schedule (experiment, off by default): same MFMA order per accumulator
        two_group = ops.basicblock64_bf16(x, w1, s1, b1, w2, s2, b2)
        torch.cuda.synchronize()
    finally:
        ops.set_tuning(ops.TUNE_BF16_FUSE_BLOCK, 3)
    assert bool(torch.isfinite(one.float()).all()) and bool(torch.isfinite(per.float()).all())
    for name, got in (("one tile per workgroup", one), ("persistent", per), ("two-group schedule", two_group)):
        bad = (got != parts).nonzero()
        assert bad.numel() == 0, (name, shape, int(bad.shape[0]), bad[:3].tolist(), bad[-3:].tolist())


def test_fused_basicblock64_refuses_shapes_it_does_not_take(dev):
    """Maps wider than 120 pixels (two strips of 60 + 2 columns: the widest whose patch rows fit 64 slots) are outside the fused kernel's patch budget: the entry point says so
    (RPG_ERR_BAD_ARG) and the composite forward falls back to two launches (same results either way)."""
    from relpose_gnn_amd import _lib, ops
    x = torch.zeros((1, 64, 123, 64), dtype=torch.bfloat16, device=dev)
    wz = torch.zeros((64, 3, 3, 64), dtype=torch.bfloat16, device=dev)
    v = torch.ones(64, device=dev)
    with pytest.raises((ValueError, _lib.RpgError)):
        ops.basicblock64_bf16(x, wz, v, v, wz, v, v)


def test_bf16_encoder_with_and_without_block_fusion_is_bit_identical(dev):
    """The bf16 model with RPG_TUNE_BF16_FUSE_BLOCK on (default) / off: 6 graphs x 8 nodes x 224x224 (layer 1: 48 images of 56x56x64
    = 294 output tiles per fused launch); the poses are equal bit for bit."""
    import relpose_gnn_amd.synth as S
    from relpose_gnn_amd import ops
    from relpose_gnn_amd.graph import fc_batch
    from relpose_gnn_amd.posenet import PoseNetX_R2
    from relpose_gnn_amd.resnet import resnet34
    D = 2048
    m = PoseNetX_R2(resnet34(), droprate=0.0, pretrained=False, feat_dim=D, edge_feat_dim=D, node_dim=D,
                    input_img_height=224, use_gnn=True, knn=-1, use_AP=True, gnn_recursion=2)
    m.load_state_dict(S.synth_state_dict(S.posenet_r2_param_shapes(D, D, D), seed=1))
    m = m.to(dev).eval()
    m.encoder_dtype = "bf16"
    d = fc_batch(S.synth_images(48, 224, 224, seed=9), 8).to(dev)
    outs = []
    try:
        for fuse in (3, 0, 1, 5):                         # persistent (default) | two launches | one tile per workgroup | two-group schedule
            ops.set_tuning(ops.TUNE_BF16_FUSE_BLOCK, fuse)
            a, r, _ = m(d)
            outs.append((a.clone(), r.clone()))
    finally:
        ops.set_tuning(ops.TUNE_BF16_FUSE_BLOCK, 3)
    for k in (1, 2, 3):
        assert torch.equal(outs[0][0], outs[k][0]) and torch.equal(outs[0][1], outs[k][1]), k
    assert bool(torch.isfinite(outs[0][0]).all()) and float(outs[0][0].abs().max()) > 0


def test_bf16_encoder_paired_downsample_launch_is_bit_identical(dev):
    """Round 6: the 3x3 / stride-2 convolution and the 1x1 / stride-2 shortcut of a down-sampling BasicBlock
    (torchvision BasicBlock.downsample, reached from modules/posenet.py:1037) as ONE launch of the LDS-DMA kernel
    (RPG_TUNE_BF16_PAIR, default on) / as two launches: same tiles, same arithmetic -- the poses of the bf16 model are equal bit for
    bit, at 224x224 (48 images: layers 2-4 all pair) and at an odd-sized 136x200 input."""
    import relpose_gnn_amd.synth as S
    from relpose_gnn_amd import ops
    from relpose_gnn_amd.graph import fc_batch
    from relpose_gnn_amd.posenet import PoseNetX_R2
    from relpose_gnn_amd.resnet import resnet34
    D = 2048
    for (hh, ww, nimg) in ((224, 224, 48), (136, 200, 64)):
        m = PoseNetX_R2(resnet34(), droprate=0.0, pretrained=False, feat_dim=D, edge_feat_dim=D, node_dim=D,
                        input_img_height=hh, use_gnn=True, knn=-1, use_AP=True, gnn_recursion=2)
        m.load_state_dict(S.synth_state_dict(S.posenet_r2_param_shapes(D, D, D), seed=1))
        m = m.to(dev).eval()
        m.encoder_dtype = "bf16"
        d = fc_batch(S.synth_images(nimg, hh, ww, seed=19), 8).to(dev)
        outs = []
        try:
            for pair in (1, 0):
                ops.set_tuning(ops.TUNE_BF16_PAIR, pair)
                a, r, _ = m(d)
                outs.append((a.clone(), r.clone()))
        finally:
            ops.set_tuning(ops.TUNE_BF16_PAIR, 1)
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), (hh, ww)
        assert bool(torch.isfinite(outs[0][0]).all()) and float(outs[0][0].abs().max()) > 0


def test_conv_bf16_lean_epilogue_without_relu_keeps_nan(dev):
    """the lean epilogue clamped with max(y, relu ? 0 : -inf), which turns a NaN accumulator of a NON-ReLU convolution
    (the 1x1 downsample) into -inf where the general epilogue and the reference propagate it.  Now a select: relu = False on
    both epilogues, one NaN pixel in the input -- the outputs agree bit for bit (NaN positions included) and the NaN is there."""
    from relpose_gnn_amd import ops
    x = _rand(24, 64, 28, 28, seed=71).bfloat16()
    x[3, 5, 6, 8] = float("nan")                          # an even pixel: the stride-2 1x1 convolution samples it
    wt = _rand(128, 64, 1, 1, seed=72, scale=(2.0 / 64) ** 0.5).bfloat16()
    scale, shift = torch.rand(128, generator=torch.Generator().manual_seed(73)) + 0.5, _rand(128, seed=74, scale=0.1)
    xd, wd = x.permute(0, 2, 3, 1).contiguous().to(dev), wt.permute(0, 2, 3, 1).contiguous().to(dev)
    outs = {}
    try:
        for lean in (1, 0):
            ops.set_tuning(ops.TUNE_BF16_LEAN_EPI, lean)
            outs[lean] = ops.conv2d_bn_act_nhwc_bf16(xd, wd, scale.to(dev), shift.to(dev), None, stride=2, pad=0, relu=False)
    finally:
        ops.set_tuning(ops.TUNE_BF16_LEAN_EPI, 1)
    a, b = outs[1].float().cpu(), outs[0].float().cpu()
    assert torch.equal(torch.isnan(a), torch.isnan(b)) and int(torch.isnan(a).sum()) == 128          # output pixel (3, 4) of image 3, all channels
    assert torch.equal(torch.nan_to_num(a, nan=0.0), torch.nan_to_num(b, nan=0.0))
    assert float(a[~torch.isnan(a)].min()) < 0.0                                                     # really no ReLU


@pytest.mark.parametrize("n,h,w,c,res", [(400, 14, 14, 256, True), (400, 28, 28, 128, False), (340, 14, 14, 256, False)])
def test_patch_kernel_tail_retiling_is_bit_identical(dev, n, h, w, c, res):
    """Round 5: on more than one round of tiles the rows beyond the last FULL round go to a second launch
    of the patch kernel with smaller tiles (RPG_TUNE_BF16_TAIL = 1: 160 x 256 on 8 x 1 waves for layer 3, 256 x 128 for layer 2)
    instead of a mostly empty round of full-size tiles.  Same K order, same MFMA, same epilogue: bit-identical to the single
    launch (= 0), with and without a residual; and within the bf16 bar of F.conv2d on a sample of images from both parts."""
    from relpose_gnn_amd import ops
    g = torch.Generator().manual_seed(n + c)
    x = torch.randn((n, h, w, c), generator=g).bfloat16().to(dev)
    wt = (torch.randn((c, 3, 3, c), generator=g) * (2.0 / (9 * c)) ** 0.5).bfloat16().to(dev)
    sc, sh = (torch.rand(c, generator=g) + 0.5).to(dev), (torch.randn(c, generator=g) * 0.1).to(dev)
    r = torch.randn((n, h, w, c), generator=g).bfloat16().to(dev) if res else None
    outs = {}
    try:
        for tail in (3, 0):                                   # both re-tilings on (layer-2 and layer-3 style) / one launch
            ops.set_tuning(ops.TUNE_BF16_TAIL, tail)
            outs[tail] = ops.conv2d_bn_act_nhwc_bf16(x, wt, sc, sh, r, stride=1, pad=1, relu=True)
    finally:
        ops.set_tuning(ops.TUNE_BF16_TAIL, 1)
    assert torch.equal(outs[3], outs[0])
    for i in (0, n // 2, n - 1):                               # first image, one in the middle, the last (tail launch)
        ref = F.conv2d(x[i:i + 1].float().permute(0, 3, 1, 2).cpu(), wt.float().permute(0, 3, 1, 2).cpu(), padding=1)
        ref = ref * sc.cpu().view(1, -1, 1, 1) + sh.cpu().view(1, -1, 1, 1)
        if res:
            ref = ref + r[i:i + 1].float().permute(0, 3, 1, 2).cpu()
        assert rel_err(outs[3][i:i + 1].float().cpu().permute(0, 3, 1, 2), F.relu(ref)) < 1e-2


def test_matrix_pipe_probe_runs_and_orders_the_operand_data(dev):
    """rpg_probe_mfma_bf16 (measurement aid of bench.py's bf16 roofline): a registers-only MFMA stream; on MI355X the sustained
    rate depends on the operand data through the power cap -- zeros fastest (2.5 PFLOP/s), ReLU-like (2.1), random (1.8).  The test
    pins what is robust: it runs, the rate is in the physical range, zeros are not slower than random data, bad arguments are refused."""
    from relpose_gnn_amd import _lib, ops
    z = ops.probe_mfma_bf16("zeros", iters=4000)
    r = ops.probe_mfma_bf16("random", iters=4000)
    assert 0.3 < r < 2.7 and 0.3 < z < 2.7, (z, r)
    assert z > 0.85 * r, (z, r)
    sink = torch.zeros(1, device=dev)
    assert _lib.lib().rpg_probe_mfma_bf16(None, 10, 256, sink.data_ptr(), None) == _lib.RPG_ERR_BAD_ARG
    assert _lib.lib().rpg_probe_mfma_bf16(sink.data_ptr(), 0, 256, sink.data_ptr(), None) == _lib.RPG_ERR_BAD_ARG
