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
    (40, 56, 56, 64, 64, 3, 1, 1, True, True, False),     # layer1 shape -> 256x64 tile
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
