"""Parity tests of the nested 2-D Winograd probe kernel (tools/probes/winograd2d.hip).  Not collected by the product test suite:
needs a library built with RPG_BUILD_DEFINES="-DRPG_PROBE_WINO2D".  Run:  python -m pytest tools/probes/test_wino2d_probe.py
(round 4: 10 passed on MI355X, profiles/r4_wino2d_nested_kernel.txt)."""
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


# ---- nested 2-D Winograd F(4x2, 3x3) (csrc/winograd2d.hip, round 4) ---------------------------------------------------------
@pytest.mark.parametrize("n,h,w,cin,cout,res,relu", [
    (2, 56, 56, 64, 64, True, True),       # ResNet34 layer 1
    (3, 28, 28, 128, 128, False, True),    # layer 2
    (5, 14, 14, 256, 256, True, True),     # layer 3
    (9, 7, 7, 512, 512, True, True),       # layer 4: odd height and width (second row / last column of border tiles masked)
    (2, 64, 86, 64, 64, True, False),      # the 256x341 evaluation shape's layer 1: ragged width 86 -> 88
    (3, 43, 32, 128, 128, True, True),     # odd height 43, layer 2 of the evaluation shape (transposed)
    (1, 2, 4, 8, 32, False, True),         # one tile, one K step
    (7, 5, 9, 24, 96, True, True),         # odd everything, Cin = 3 K steps, Cout = 3 channel tiles
    (130, 8, 8, 16, 32, True, True),       # many images per workgroup (64 tiles span 8 images)
])
def test_conv3x3_winograd_nested_2d(dev, n, h, w, cin, cout, res, relu):
    """The nested Winograd F(4x2, 3x3) kernel (RPG_TUNE_WINO2D = 2: wherever eligible) against F.conv2d on the CPU, 2e-5 (the bar
    of the 1-D kernel), and against the 1-D kernel on the same operands.  Reference op: torchvision BasicBlock conv + bn (+
    identity) + relu reached from posenet.py:1037."""
    from relpose_gnn_amd import ops
    g = torch.Generator().manual_seed(4000 + h * w + cin)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    r = torch.randn(n, cout, h, w, generator=g) if res else None
    ref = F.conv2d(x, wt, None, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    if res:
        ref = ref + r
    if relu:
        ref = F.relu(ref)
    nh = lambda t: None if t is None else t.permute(0, 2, 3, 1).contiguous().to(dev)
    u = ops.wino43_transform_weights(nh(wt))
    assert u.numel() == 42 * cout * cin, "library built without -DRPG_PROBE_WINO2D"
    xd, rd, sc, sh = nh(x), nh(r), scale.to(dev), shift.to(dev)
    ops.set_tuning(ops.TUNE_WINOGRAD, 3)
    try:
        ops.set_tuning(ops.TUNE_WINO2D, 2)
        y2 = ops.conv3x3_wino43_bn_act_nhwc(xd, u, sc, sh, rd, relu=relu)
        y2b = ops.conv3x3_wino43_bn_act_nhwc(xd, u, sc, sh, rd, relu=relu)
        ops.set_tuning(ops.TUNE_WINO2D, 0)
        y1 = ops.conv3x3_wino43_bn_act_nhwc(xd, u, sc, sh, rd, relu=relu)
    finally:
        ops.set_tuning(ops.TUNE_WINO2D, 1)
        ops.set_tuning(ops.TUNE_WINOGRAD, 1)
    e2, e1 = rel_err(y2.cpu().permute(0, 3, 1, 2), ref), rel_err(y1.cpu().permute(0, 3, 1, 2), ref)
    assert torch.equal(y2, y2b)
    assert e1 < 2e-5 and e2 < 2e-5, (e1, e2)


def test_winograd_nested_2d_falls_back_when_not_eligible(dev):
    """Cout % 32 != 0 or Cin % 8 != 0: the launcher keeps the 1-D kernel even with RPG_TUNE_WINO2D = 2 (same bits as = 0)."""
    from relpose_gnn_amd import ops
    g = torch.Generator().manual_seed(77)
    for cin, cout in ((12, 32), (16, 40)):
        x = torch.randn(2, 9, 11, cin, generator=g).to(dev)
        wt = (torch.randn(cout, 3, 3, cin, generator=g) * 0.1).to(dev)
        u = ops.wino43_transform_weights(wt)
        outs = []
        for mode in (2, 0):
            ops.set_tuning(ops.TUNE_WINO2D, mode)
            ops.set_tuning(ops.TUNE_WINOGRAD, 3)
            try:
                outs.append(ops.conv3x3_wino43_bn_act_nhwc(x, u, None, None, None, relu=False))
            finally:
                ops.set_tuning(ops.TUNE_WINO2D, 1)
                ops.set_tuning(ops.TUNE_WINOGRAD, 1)
        assert torch.equal(outs[0], outs[1])
