"""Seeded random-shape sweeps of the tile engine front-ends against PyTorch-CPU fp32: odd kernel sizes, strides, paddings,
channel counts that are not tile multiples, ragged M / N / K, every residual / ReLU / scale combination.  Complements the
hand-picked shapes of test_hip_ops.py (which cover the ResNet34 / GNN layer shapes)."""
import random

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


@pytest.mark.parametrize("case", range(24))
def test_random_conv_direct(dev, case):
    from relpose_gnn_amd import ops
    rng = random.Random(1000 + case)
    kh, kw = rng.choice([1, 2, 3, 5, 7]), rng.choice([1, 2, 3, 5])
    stride, pad = rng.choice([1, 2, 3]), rng.choice([0, 1, 2, 3])
    cin, cout = 4 * rng.randint(1, 24), rng.choice([4, 12, 32, 60, 64, 100, 128, 200])
    n, h, w = rng.randint(1, 5), rng.randint(max(kh - 2 * pad, 1), 30), rng.randint(max(kw - 2 * pad, 1), 30)
    if (h + 2 * pad - kh) < 0 or (w + 2 * pad - kw) < 0:
        pytest.skip("empty output")
    res, relu, has_scale = rng.random() < 0.5, rng.random() < 0.5, rng.random() < 0.7
    x = _rand(n, cin, h, w, seed=case)
    wt = _rand(cout, cin, kh, kw, seed=case + 1, scale=(1.0 / (cin * kh * kw)) ** 0.5)
    scale = (torch.rand(cout, generator=torch.Generator().manual_seed(case + 2)) + 0.5) if has_scale else None
    shift = _rand(cout, seed=case + 3, scale=0.2)
    ref = F.conv2d(x, wt, None, stride=stride, padding=pad)
    if scale is not None:
        ref = ref * scale.view(1, -1, 1, 1)
    ref = ref + shift.view(1, -1, 1, 1)
    r = None
    if res:
        r = _rand(*ref.shape, seed=case + 4)
        ref = ref + r
    if relu:
        ref = F.relu(ref)
    y = ops.conv2d_bn_act_nhwc(x.permute(0, 2, 3, 1).contiguous().to(dev), wt.permute(0, 2, 3, 1).contiguous().to(dev),
                               None if scale is None else scale.to(dev), shift.to(dev),
                               None if r is None else r.permute(0, 2, 3, 1).contiguous().to(dev), stride=stride, pad=pad, relu=relu)
    assert rel_err(y.cpu().permute(0, 3, 1, 2), ref) < 1e-5, (kh, kw, stride, pad, cin, cout, n, h, w)


@pytest.mark.parametrize("case", range(24))
def test_random_conv_buffer_loaders(dev, case):
    """Shapes that take the buffer-load loaders of the tile engine: Cin a multiple of 16 / 32 (ConvLoaderB) or Cin = 4 with
    a >= 4-wide kernel (ConvLoaderTap), random kernel / stride / padding / image sizes (many images per tile, rows that
    start in the padding, K tails for the tap loader)."""
    from relpose_gnn_amd import ops
    rng = random.Random(5000 + case)
    if case % 3 == 2:
        cin, kh, kw = 4, rng.choice([3, 5, 7]), rng.choice([4, 5, 7])
    else:
        cin, kh, kw = rng.choice([16, 32, 48, 64, 96, 128]), rng.choice([1, 2, 3]), rng.choice([1, 2, 3])
    stride, pad = rng.choice([1, 2, 3]), rng.choice([0, 1, 2, 3])
    cout = rng.choice([4, 32, 60, 64, 100, 128, 200])
    n, h, w = rng.randint(1, 9), rng.randint(max(kh - 2 * pad, 1), 30), rng.randint(max(kw - 2 * pad, 1), 30)
    res, relu = rng.random() < 0.5, rng.random() < 0.5
    x = _rand(n, cin, h, w, seed=case)
    wt = _rand(cout, cin, kh, kw, seed=case + 1, scale=(1.0 / (cin * kh * kw)) ** 0.5)
    scale = torch.rand(cout, generator=torch.Generator().manual_seed(case + 2)) + 0.5
    shift = _rand(cout, seed=case + 3, scale=0.2)
    ref = F.conv2d(x, wt, None, stride=stride, padding=pad) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    r = None
    if res:
        r = _rand(*ref.shape, seed=case + 4)
        ref = ref + r
    if relu:
        ref = F.relu(ref)
    y = ops.conv2d_bn_act_nhwc(x.permute(0, 2, 3, 1).contiguous().to(dev), wt.permute(0, 2, 3, 1).contiguous().to(dev),
                               scale.to(dev), shift.to(dev), None if r is None else r.permute(0, 2, 3, 1).contiguous().to(dev),
                               stride=stride, pad=pad, relu=relu)
    assert rel_err(y.cpu().permute(0, 3, 1, 2), ref) < 1e-5, (kh, kw, stride, pad, cin, cout, n, h, w)


@pytest.mark.parametrize("case", range(12))
def test_random_linear_aligned(dev, case):
    """Ungathered sources whose widths are multiples of 32: GatherLoaderB + the interleaved main loop (all tile shapes
    the size heuristics pick, stream-K included)."""
    from relpose_gnn_amd import ops
    rng = random.Random(6000 + case)
    ns = rng.randint(1, 3)
    widths = [32 * rng.randint(1, 20) for _ in range(ns)]
    m, n_out = rng.randint(1, 2000), 4 * rng.randint(1, 600)
    srcs = [(_rand(m, wd, seed=case * 5 + i).to(dev), None) for i, wd in enumerate(widths)]
    k = sum(widths)
    wl, bias = _rand(n_out, k, seed=case + 60, scale=k ** -0.5), _rand(n_out, seed=case + 61)
    res = _rand(m, n_out, seed=case + 62) if rng.random() < 0.5 else None
    relu = rng.random() < 0.5
    ref = F.linear(torch.cat([a.cpu() for a, _ in srcs], 1), wl, bias)
    if res is not None:
        ref = ref + res
    if relu:
        ref = F.relu(ref)
    y = ops.linear_gather(srcs, wl.to(dev), bias.to(dev), m, None if res is None else res.to(dev), relu)
    assert rel_err(y.cpu(), ref) < 1e-5, (widths, m, n_out)


@pytest.mark.parametrize("kernel", [2, 3], ids=["wave4", "wave8"])
@pytest.mark.parametrize("case", range(16))
def test_random_conv_winograd(dev, case, kernel):
    from relpose_gnn_amd import ops
    rng = random.Random(2000 + case)
    cin, cout = 4 * rng.randint(1, 40), 4 * rng.randint(1, 50)
    n, h, w = rng.randint(1, 6), rng.randint(1, 20), rng.randint(1, 37)
    res, relu = rng.random() < 0.5, rng.random() < 0.5
    x = _rand(n, cin, h, w, seed=case)
    wt = _rand(cout, cin, 3, 3, seed=case + 1, scale=(1.0 / (cin * 9)) ** 0.5)
    scale = torch.rand(cout, generator=torch.Generator().manual_seed(case + 2)) + 0.5
    shift = _rand(cout, seed=case + 3, scale=0.2)
    ref = F.conv2d(x, wt, None, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    r = None
    if res:
        r = _rand(*ref.shape, seed=case + 4)
        ref = ref + r
    if relu:
        ref = F.relu(ref)
    u = ops.wino43_transform_weights(wt.permute(0, 2, 3, 1).contiguous().to(dev))
    ops.set_tuning(ops.TUNE_WINOGRAD, kernel)          # both Winograd kernels on every shape (1 = choose by size)
    try:
        y = ops.conv3x3_wino43_bn_act_nhwc(x.permute(0, 2, 3, 1).contiguous().to(dev), u, scale.to(dev), shift.to(dev),
                                           None if r is None else r.permute(0, 2, 3, 1).contiguous().to(dev), relu=relu)
    finally:
        ops.set_tuning(ops.TUNE_WINOGRAD, 1)
    assert rel_err(y.cpu().permute(0, 3, 1, 2), ref) < 2e-5, (cin, cout, n, h, w)


@pytest.mark.parametrize("case", range(20))
def test_random_linear_gather(dev, case):
    from relpose_gnn_amd import ops
    rng = random.Random(3000 + case)
    ns = rng.randint(1, 3)
    widths = [4 * rng.randint(1, 90) for _ in range(ns)]
    m, n_out, rows = rng.randint(1, 700), rng.randint(1, 300), rng.randint(1, 60)
    g0 = torch.Generator().manual_seed(case)
    srcs, cat = [], []
    for i, wd in enumerate(widths):
        if rng.random() < 0.6:
            a = _rand(rows, wd, seed=case * 7 + i)
            idx = torch.randint(0, rows, (m,), generator=g0)
            srcs.append((a.to(dev), idx.to(dev)))
            cat.append(a[idx])
        else:
            a = _rand(m, wd, seed=case * 7 + i)
            srcs.append((a.to(dev), None))
            cat.append(a)
    k = sum(widths)
    wl, bias = _rand(n_out, k, seed=case + 50, scale=k ** -0.5), _rand(n_out, seed=case + 51)
    res = _rand(m, n_out, seed=case + 52) if rng.random() < 0.5 else None
    relu = rng.random() < 0.5
    ref = F.linear(torch.cat(cat, 1), wl, bias)
    if res is not None:
        ref = ref + res
    if relu:
        ref = F.relu(ref)
    y = ops.linear_gather(srcs, wl.to(dev), bias.to(dev), m, None if res is None else res.to(dev), relu)
    assert rel_err(y.cpu(), ref) < 1e-5, (widths, m, n_out)


@pytest.mark.parametrize("case", range(12))
def test_random_conv_bf16(dev, case):
    from relpose_gnn_amd import ops
    rng = random.Random(4000 + case)
    kh = kw = rng.choice([1, 3, 5])
    stride, pad = rng.choice([1, 2]), rng.choice([0, 1, 2])
    cin, cout = 8 * rng.randint(1, 20), 4 * rng.randint(1, 40)
    n, h, w = rng.randint(1, 4), rng.randint(kh, 24), rng.randint(kw, 24)
    res, relu = rng.random() < 0.5, rng.random() < 0.5
    x = _rand(n, cin, h, w, seed=case).bfloat16()
    wt = _rand(cout, cin, kh, kw, seed=case + 1, scale=(1.0 / (cin * kh * kw)) ** 0.5).bfloat16()
    scale = torch.rand(cout, generator=torch.Generator().manual_seed(case + 2)) + 0.5
    shift = _rand(cout, seed=case + 3, scale=0.2)
    ref = F.conv2d(x.float(), wt.float(), None, stride=stride, padding=pad) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    r = None
    if res:
        r = _rand(*ref.shape, seed=case + 4).bfloat16()
        ref = ref + r.float()
    if relu:
        ref = F.relu(ref)
    y = ops.conv2d_bn_act_nhwc_bf16(x.permute(0, 2, 3, 1).contiguous().to(dev), wt.permute(0, 2, 3, 1).contiguous().to(dev),
                                    scale.to(dev), shift.to(dev), None if r is None else r.permute(0, 2, 3, 1).contiguous().to(dev),
                                    stride=stride, pad=pad, relu=relu)
    assert rel_err(y.float().cpu().permute(0, 3, 1, 2), ref) < 1e-2, (kh, stride, pad, cin, cout, n, h, w)


@pytest.mark.parametrize("case", range(20))
def test_random_stem_bf16_strips(dev, case):
    """Round 6: the strip-march stem (csrc/stem_bf16.hip; torchvision conv1 / bn1 / relu / maxpool reached from posenet.py:1037) on
    random image counts and sizes -- ragged strips (pooled widths that are no multiple of 15), ragged bands, odd convolution heights
    and widths, images narrower than one strip, odd image sizes with bf16 input (rows and images that start 2 bytes past a dword)
    -- in every variant of the kernel, fp32 and host-rounded bf16 input (bit-identical), against conv2d + BN + ReLU + max_pool2d
    on the same bf16-rounded operands."""
    from relpose_gnn_amd import ops
    from relpose_gnn_amd.params import pack_stem_bf16
    rng = random.Random(4000 + case)
    n, h, w = rng.randint(1, 9), rng.randint(1, 150), rng.randint(1, 260)
    variant = rng.choice([1, 33]) + (rng.choice([0, 3, 7, 11]) << 8)
    x = _rand(n, 3, h, w, seed=case)
    wt = _rand(64, 3, 7, 7, seed=case + 1, scale=(2.0 / 147) ** 0.5)
    g = torch.Generator().manual_seed(case + 2)
    scale = (torch.rand(64, generator=g) + 0.5) * torch.where(torch.rand(64, generator=g) < 0.2, -1.0, 1.0)
    shift = _rand(64, seed=case + 3, scale=0.3)
    conv = F.conv2d(x.bfloat16().float(), wt.bfloat16().float(), None, stride=2, padding=3)
    ref = F.max_pool2d(F.relu(conv * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)), 3, 2, 1).bfloat16().float()
    ops.set_tuning(ops.TUNE_FUSED_STEM, variant)
    try:
        wp = pack_stem_bf16(wt).to(dev)
        y = ops.stem_conv_bn_relu_maxpool_bf16(x.to(dev), wp, scale.to(dev), shift.to(dev))
        y16 = ops.stem_conv_bn_relu_maxpool_bf16(x.bfloat16().to(dev), wp, scale.to(dev), shift.to(dev))
    finally:
        ops.set_tuning(ops.TUNE_FUSED_STEM, 1)
    got = y.float().cpu().permute(0, 3, 1, 2)
    assert got.shape == ref.shape and torch.equal(y, y16), (n, h, w, variant)
    assert rel_err(got, ref) < 1e-2, (n, h, w, variant)
    assert float((got - ref).abs().mean() / ref.abs().mean().clamp(min=1e-30)) < 3e-4, (n, h, w, variant)


@pytest.mark.parametrize("case", range(16))
def test_random_fused_basicblock64(dev, case):
    """Round 6: the fused 64-channel BasicBlock (csrc/block_bf16.inc) on random maps 4..120 pixels wide -- one strip up to 62, two
    column strips beyond --, random heights and image counts (tiles straddling images and strips), against the block in fp32 on
    the same bf16 operands with the intermediate rounded to bf16 (the bf16 bar of the convolution tests); shapes the kernel does not
    take must say so (ValueError), never give wrong numbers."""
    from relpose_gnn_amd import ops
    rng = random.Random(5000 + case)
    n, h, w = rng.randint(1, 6), rng.randint(1, 70), rng.randint(4, 120)
    g = torch.Generator().manual_seed(case)
    x = torch.randn((n, h, w, 64), generator=g).bfloat16()
    w1 = (torch.randn((64, 3, 3, 64), generator=g) * (2.0 / 576) ** 0.5).bfloat16()
    w2 = (torch.randn((64, 3, 3, 64), generator=g) * (2.0 / 576) ** 0.5).bfloat16()
    s1, b1 = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.2
    s2, b2 = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.2
    try:
        got = ops.basicblock64_bf16(x.to(dev), w1.to(dev), s1.to(dev), b1.to(dev), w2.to(dev), s2.to(dev), b2.to(dev))
    except ValueError:
        pytest.skip(f"shape {(n, h, w)} not taken by the fused kernel (LDS budget): the encoder uses two launches")
    xf = x.float().permute(0, 3, 1, 2)
    tf = torch.relu(F.conv2d(xf, w1.float().permute(0, 3, 1, 2), padding=1) * s1.view(1, -1, 1, 1) + b1.view(1, -1, 1, 1)).bfloat16().float()
    yf = torch.relu(F.conv2d(tf, w2.float().permute(0, 3, 1, 2), padding=1) * s2.view(1, -1, 1, 1) + b2.view(1, -1, 1, 1) + xf)
    gotf = got.float().cpu().permute(0, 3, 1, 2)
    assert rel_err(gotf, yf) < 1e-2, (n, h, w)
    assert float((gotf - yf).abs().mean() / yf.abs().mean().clamp(min=1e-30)) < 3e-3, (n, h, w)
