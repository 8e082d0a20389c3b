"""Parity test of the weights-stationary probe kernel (tools/probes/conv3x3_bf16_ws64.inc).  Not collected by the product test
suite: needs a library built with RPG_BUILD_DEFINES="-DRPG_PROBE_WS64".  Run:  python -m pytest tools/probes/test_ws64_probe.py"""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from conftest import rel_err  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _rand(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


@pytest.mark.parametrize("mode", [1, 2], ids=["tile384", "tile256"])
@pytest.mark.parametrize("n,h,w,res,relu", [
    (40, 56, 56, True, True),        # layer-1 shape at 224 x 224: 64 slots per patch row, tiles cross image rows and images
    (3, 56, 56, False, True),        # 9408 pixels: fewer tiles than CUs, ragged last tile
    (4, 64, 86, True, True),         # layer 1 of 256 x 341 images: 96 slots per row
    (37, 17, 23, True, False),       # odd sizes: 32 slots per row, tiles span two to three images (zero rows between them)
    (700, 56, 56, True, True),       # 5717 / 8575 tiles: every workgroup walks 23-34 tiles, both patch buffers, prefetch and deferred stores across tiles
])
def test_conv3x3_bf16_weights_stationary_kernel(dev, mode, n, h, w, res, relu):
    """The weights-stationary kernel of the bf16 encoder's 64 -> 64 channel 3x3 convolutions (RPG_TUNE_BF16_WS64 = 1: 384-pixel
    tiles, 2: 256; all weights in registers, persistent 4-wave workgroups, patch of the next (tile, chunk) fetched during the
    current one) against F.conv2d on the same bf16 inputs in fp32 (ResNet layer 1, torchvision BasicBlock conv1 / conv2 reached
    from posenet.py:1037).  Each output depends on all 576 weights and 9 patch slots: a wrong register, slot or swizzle moves it."""
    from relpose_gnn_amd import ops
    x = _rand(n, 64, h, w, seed=41).bfloat16()
    wt = _rand(64, 64, 3, 3, seed=42, scale=(2.0 / 576) ** 0.5).bfloat16()
    scale = torch.rand(64, generator=torch.Generator().manual_seed(43)) + 0.5
    shift = _rand(64, seed=44, scale=0.1)
    ref = F.conv2d(x.float(), wt.float(), None, stride=1, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    r = None
    if res:
        r = _rand(*ref.shape, seed=45).bfloat16()
        ref = ref + r.float()
    if relu:
        ref = F.relu(ref)
    xd, wd = x.permute(0, 2, 3, 1).contiguous().to(dev), wt.permute(0, 2, 3, 1).contiguous().to(dev)
    rd = None if r is None else r.permute(0, 2, 3, 1).contiguous().to(dev)
    ops.set_tuning(ops.TUNE_BF16_WS64, mode)
    try:
        y = ops.conv2d_bn_act_nhwc_bf16(xd, wd, scale.to(dev), shift.to(dev), rd, stride=1, pad=1, relu=relu)
        ops.set_tuning(ops.TUNE_BF16_WS64, 0)
        y0 = ops.conv2d_bn_act_nhwc_bf16(xd, wd, scale.to(dev), shift.to(dev), rd, stride=1, pad=1, relu=relu)
    finally:
        ops.set_tuning(ops.TUNE_BF16_WS64, 0)
    got = y.float().cpu().permute(0, 3, 1, 2)
    assert rel_err(got, ref) < 1e-2
    assert float((got - ref).abs().mean() / ref.abs().mean().clamp(min=1e-30)) < 3e-3     # bf16 output rounding only
    # against the kernel it replaces: same products, different summation order -> at most a bf16 ulp on single elements
    assert rel_err(y.float().cpu(), y0.float().cpu()) < 1e-2


