"""Parity at the launch geometry that bench.py TIMES ("What's weak" 1-2).

The per-op tests elsewhere use small problems (<= 49 Winograd workgroups); at BASELINE.json configs[1] every Winograd
layer is a grid of 392-1568 workgroups = full rounds of wino43_conv8_kernel<false> + a split-K <true> tail + the fix-up
kernel in ONE call, the XCD remap runs over > 256 ids, a buffer resource spans 128-256 images, and the batch is spread
over two HIP streams.  These tests run exactly that:
  * the configs[1] forward as bench.py builds it (32 graphs x 8 nodes x 224x224, D=2048, hip_streams=2) against the CPU
    oracle on all 32 graphs, <= 1e-4 (the north-star bar), also through a PyG-style batch without `graph_sizes`;
  * rpg_conv3x3_wino43_bn_act_nhwc_f32 at the four ResNet34 stage shapes with 128 and 256 images vs F.conv2d, 2e-5;
  * the bf16 encoder (and bf16 GNN) at the 256x341 evaluation shape, configs[4]'s dtype x shape, one graph;
  * the seeded differential fuzz of tools/fuzz_kernels.py, against F.conv2d (CPU fp32) instead of a sibling kernel.
Oracle: oracle/posenet_ref.py (restates /root/reference/python/niantic/modules/posenet.py:1033-1091; checker only).
"""
import json
import os
import random
import types

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _report(rec):
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "parity_report.jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")


def _r3_model(dev, img_h=224):
    import relpose_gnn_amd.synth as S
    from relpose_gnn_amd.posenet import PoseNetX_R2
    from relpose_gnn_amd.resnet import resnet34
    D = 2048
    m = PoseNetX_R2(resnet34(), droprate=0.0, pretrained=False, feat_dim=D, edge_feat_dim=D, node_dim=D,
                    input_img_height=img_h, use_gnn=True, knn=-1, use_AP=True, gnn_recursion=2)
    sd = S.synth_state_dict(S.posenet_r2_param_shapes(D, D, D), seed=1)
    m.load_state_dict(sd)
    return m.to(dev).eval(), sd


def test_configs1_forward_as_benched_vs_oracle(dev):
    """BASELINE.json configs[1] exactly as bench.py runs it: 32 graphs x 8 nodes, 224x224, fp32, two streams."""
    from oracle import posenet_ref as O
    from relpose_gnn_amd.graph import fc_batch
    m, sd = _r3_model(dev)
    G, N = 32, 8
    x = torch.randn((G * N, 3 * 224 * 224), generator=torch.Generator().manual_seed(4321))
    data = fc_batch(x, N).to(dev)
    m.hip_streams = 2
    a2, r2, ei = m(data)
    m.check_edge_index()
    assert ei is data.edge_index and a2.shape == (G * N, 6) and r2.shape == (G * 56, 6)
    # the CPU oracle, 4 graphs at a time (graphs are independent; bounds the host memory)
    oa, orr = [], []
    e1 = O.batch_edge_index(N, 4)
    for g0 in range(0, G, 4):
        a, r, _ = O.posenet_forward(sd, x[g0 * N:(g0 + 4) * N], e1, 224, 2)
        oa.append(a)
        orr.append(r)
    oa, orr = torch.cat(oa), torch.cat(orr)
    ea, er = rel_err(a2.cpu(), oa), rel_err(r2.cpu(), orr)
    # per-graph worst case too: a wrong tile anywhere must not hide behind the batch-wide maximum
    pg = max(rel_err(r2[g * 56:(g + 1) * 56].cpu(), orr[g * 56:(g + 1) * 56]) for g in range(G))
    _report({"case": "configs1_32graphs_224px_2streams_vs_live_oracle", "abs_pose_rel_err": ea, "rel_pose_rel_err": er,
             "worst_graph_rel_pose_rel_err": pg})
    assert ea < 1e-4 and er < 1e-4 and pg < 2e-4, (ea, er, pg)
    # one stream (256 images per launch: the geometry of the roofline pass)
    m.hip_streams = 1
    a1, r1, _ = m(data)
    ea1, er1 = rel_err(a1.cpu(), oa), rel_err(r1.cpu(), orr)
    _report({"case": "configs1_32graphs_224px_1stream_vs_live_oracle", "abs_pose_rel_err": ea1, "rel_pose_rel_err": er1})
    assert ea1 < 1e-4 and er1 < 1e-4, (ea1, er1)
    # a torch_geometric-style Batch (collation tables, no graph_sizes) takes the same two-stream path: identical bits
    m.hip_streams = 2
    cum_n, cum_e = torch.arange(G + 1) * N, torch.arange(G + 1) * 56
    pyg = types.SimpleNamespace(x=data.x, edge_index=data.edge_index, edge_attr=None, batch=data.batch, num_graphs=G,
                                _slice_dict={"x": cum_n, "edge_index": cum_e})
    assert m._partition(pyg, G * N, G * 56) == [(0, 128, 0, 896, 0), (128, 256, 896, 1792, 1)]
    ap, rp, _ = m(pyg)
    assert torch.equal(ap, a2) and torch.equal(rp, r2)


@pytest.mark.parametrize("n", [128, 256])
@pytest.mark.parametrize("h,c", [(56, 64), (28, 128), (14, 256), (7, 512)])
def test_winograd_at_bench_grids(dev, n, h, c):
    """The four 3x3/stride-1 stage shapes of ResNet34 at 224x224 with the image counts one launch of the bench sees (128
    per stream, 256 in the roofline pass): main kernel + split-K tail + fix-up in one call, with residual."""
    from relpose_gnn_amd import ops
    g = torch.Generator().manual_seed(100 + h)
    x = torch.randn(n, c, h, h, generator=g)
    wt = torch.randn(c, c, 3, 3, generator=g) * (2.0 / (c * 9)) ** 0.5
    scale, shift = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1
    r = torch.randn(n, c, h, h, generator=g)
    ref = F.relu(F.conv2d(x, wt, None, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + r)
    u = ops.wino43_transform_weights(wt.permute(0, 2, 3, 1).contiguous().to(dev))
    xd, rd = x.permute(0, 2, 3, 1).contiguous().to(dev), r.permute(0, 2, 3, 1).contiguous().to(dev)
    tiles = (n * h * ((h + 3) // 4) + 127) // 128 * ((c + 63) // 64)
    # 784 / 1568 (layer 1), 392 / 784 (layer 2): full rounds of the 256 CUs + a split-K tail + the fix-up in one call;
    # 224 / 448 (layer 3) and 112 / 224 (layer 4): partial rounds, XCD remap over a ragged id range
    assert tiles == {(56, 128): 784, (56, 256): 1568, (28, 128): 392, (28, 256): 784, (14, 128): 224, (14, 256): 448,
                     (7, 128): 112, (7, 256): 224}[(h, n)]
    errs = {}
    for persist in (2, 1, 0):            # the persistent kernel always / where there are more tiles than CUs (default) / never
        for split in (1, 0):
            ops.set_tuning(ops.TUNE_WINO_SPLIT, split)
            ops.set_tuning(ops.TUNE_WINO_PERSIST, persist)
            try:
                y = ops.conv3x3_wino43_bn_act_nhwc(xd, u, scale.to(dev), shift.to(dev), rd, relu=True)
            finally:
                ops.set_tuning(ops.TUNE_WINO_SPLIT, 1)
                ops.set_tuning(ops.TUNE_WINO_PERSIST, 1)
            errs[(persist, split)] = rel_err(y.cpu().permute(0, 3, 1, 2), ref)
    _report({"case": f"wino43_stage_{h}x{h}x{c}_n{n}", "workgroups": tiles, "rel_err_split": errs[(1, 1)],
             "rel_err_nosplit": errs[(1, 0)], "rel_err_one_workgroup_per_tile": errs[(0, 1)],
             "rel_err_persistent_always": errs[(2, 1)]})
    assert max(errs.values()) < 2e-5, errs


# bf16 bars: stated on what the bf16 kernels control.  A chain of L bf16-input GEMM layers with
# fp32 accumulation adds ~0.5 * 2^-9 relative error per layer from rounding activations and weights (independent,
# random sign): sqrt(36) * 1e-3 = 6e-3 for the encoder -> bar 1e-2 on the features.  The randomly initialised GNN then
# amplifies a feature perturbation ~4x into the relative poses (fp64 probe, profiles/r1_parity_probe_fp64.txt) -> 2e-2;
# the abs-pose head amplifies ~9x (a property of the random head, not of a kernel) and is reported, bounded at 5e-2.
BF16_FEAT, BF16_REL, BF16_ABS = 1e-2, 2e-2, 5e-2


def test_bf16_eval_shape_256x341_vs_oracle(dev):
    """configs[4]'s dtype x shape: one 8-node graph of 256x341 images (the 7-Scenes evaluation shape,
    dataset_7Scenes_multi.py:341,434), bf16 encoder with the fp32 GNN, then with the bf16 GNN Linears as well."""
    import relpose_gnn_amd.synth as S
    from oracle import posenet_ref as O
    from relpose_gnn_amd.graph import fc_batch
    m, sd = _r3_model(dev, img_h=256)
    x = S.synth_images(8, 256, 341, seed=9)
    d = fc_batch(x, 8).to(dev)
    st = {}
    oa, orr, _ = O.posenet_forward(sd, x, d.edge_index.cpu(), 256, 2, st)
    m.encoder_dtype = "bf16"
    feat = m._enc.run(m.feature_extractor.state_dict, "", x.view(8, 3, 256, 341).to(dev))
    ef = rel_err(feat.cpu(), st["fc"])
    a, r, _ = m(d)
    ea, er = rel_err(a.cpu(), oa), rel_err(r.cpu(), orr)
    m.gnn_dtype = "bf16"
    ab, rb, _ = m(d)
    eab, erb = rel_err(ab.cpu(), oa), rel_err(rb.cpu(), orr)
    _report({"case": "bf16_256x341_1x8node_vs_fp32_oracle", "feat_rel_err": ef, "abs_pose_rel_err": ea, "rel_pose_rel_err": er,
             "bf16_gnn_abs_pose_rel_err": eab, "bf16_gnn_rel_pose_rel_err": erb})
    assert ef < BF16_FEAT and er < BF16_REL and ea < BF16_ABS, (ef, ea, er)
    assert erb < 1.5 * BF16_REL and eab < BF16_ABS, (eab, erb)     # ~20 more chained bf16-input GEMMs
    m.encoder_dtype, m.gnn_dtype = "f32", "f32"
    a32, r32, _ = m(d)
    assert rel_err(a32.cpu(), oa) < 1e-4 and rel_err(r32.cpu(), orr) < 1e-4


def _conv_case(rng, case, wino):
    if wino:
        cin, cout = 4 * rng.randint(1, 48), 4 * rng.randint(1, 80)
        kh = kw = 3
        stride, pad = 1, 1
        n, h, w = rng.randint(1, 40), rng.randint(1, 40), rng.randint(1, 60)
    else:
        if case % 4 == 3:
            cin, kh, kw = 4, rng.choice([1, 3, 5, 7]), rng.choice([4, 5, 6, 7])
        else:
            cin, kh, kw = 16 * rng.randint(1, 12), rng.choice([1, 2, 3, 4]), rng.choice([1, 2, 3])
        stride, pad = rng.choice([1, 2, 3]), rng.choice([0, 1, 2, 3])
        cout = 4 * rng.randint(1, 70)
        n, h, w = rng.randint(1, 30), rng.randint(max(kh - 2 * pad, 1), 40), rng.randint(max(kw - 2 * pad, 1), 40)
    g = torch.Generator().manual_seed(7000 + case)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, kh, kw, generator=g) * (1.0 / (cin * kh * kw)) ** 0.5
    sc, sh = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.2
    ref = F.conv2d(x, wt, None, stride=stride, padding=pad) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    r = torch.randn(ref.shape, generator=g) if rng.random() < 0.5 else None
    relu = rng.random() < 0.5
    if r is not None:
        ref = ref + r
    if relu:
        ref = F.relu(ref)
    return x, wt, sc, sh, r, relu, stride, pad, ref


@pytest.mark.parametrize("block", range(6))
def test_fuzz_winograd_vs_conv2d(dev, block):
    """Seeded random shapes (n <= 40, h <= 40, w <= 60, Cin <= 192, Cout <= 320), both Winograd kernels, split-K tail on
    and off, against F.conv2d on the CPU (16 shapes per block, 96 in all)."""
    from relpose_gnn_amd import ops
    rng = random.Random(12345 + block)
    bad = []
    for case in range(16 * block, 16 * block + 16):
        x, wt, sc, sh, r, relu, _, _, ref = _conv_case(rng, case, True)
        nh = lambda t: None if t is None else t.permute(0, 2, 3, 1).contiguous().to(dev)
        u = ops.wino43_transform_weights(nh(wt))
        for kern in (2, 3, 5):           # 5 = the 8-wave kernel without its persistent form
            for split in (0, 1):
                ops.set_tuning(ops.TUNE_WINOGRAD, min(kern, 3))
                ops.set_tuning(ops.TUNE_WINO_SPLIT, split)
                ops.set_tuning(ops.TUNE_WINO_PERSIST, int(kern != 5))
                try:
                    y = ops.conv3x3_wino43_bn_act_nhwc(nh(x), u, sc.to(dev), sh.to(dev), nh(r), relu=relu)
                finally:
                    ops.set_tuning(ops.TUNE_WINOGRAD, 1)
                    ops.set_tuning(ops.TUNE_WINO_SPLIT, 1)
                    ops.set_tuning(ops.TUNE_WINO_PERSIST, 1)
                e = rel_err(y.cpu().permute(0, 3, 1, 2), ref)
                if not e < 2e-5:
                    bad.append((case, kern, split, tuple(x.shape), wt.shape[0], e))
    assert not bad, bad


@pytest.mark.parametrize("block", range(4))
def test_fuzz_tile_engine_vs_conv2d(dev, block):
    """Seeded random convolutions (strides 1-3, paddings 0-3, 4-channel tap-loader cases) over random tile / K-step /
    stream-K / loader choices of the f32 tile engine, against F.conv2d on the CPU (16 shapes per block)."""
    from relpose_gnn_amd import ops
    rng = random.Random(777 + block)
    bad = []
    for case in range(16 * block, 16 * block + 16):
        x, wt, sc, sh, r, relu, stride, pad, ref = _conv_case(rng, case, False)
        nh = lambda t: None if t is None else t.permute(0, 2, 3, 1).contiguous().to(dev)
        tile, bk, sk, fast = rng.choice([-1, 0, 1, 2, 3]), rng.choice([0, 16, 32]), rng.choice([0, 1]), rng.choice([0, 1])
        for k, v in ((ops.TUNE_TILE, tile), (ops.TUNE_BK, bk), (ops.TUNE_STREAMK, sk), (ops.TUNE_FAST_LOADER, fast)):
            ops.set_tuning(k, v)
        try:
            y = ops.conv2d_bn_act_nhwc(nh(x), nh(wt), sc.to(dev), sh.to(dev), nh(r), stride=stride, pad=pad, relu=relu)
        finally:
            for k, v in ((ops.TUNE_TILE, -1), (ops.TUNE_BK, 0), (ops.TUNE_STREAMK, 1), (ops.TUNE_FAST_LOADER, 1)):
                ops.set_tuning(k, v)
        e = rel_err(y.cpu().permute(0, 3, 1, 2), ref)
        if not e < 1e-5:
            bad.append((case, tuple(x.shape), tuple(wt.shape), stride, pad, tile, bk, sk, fast, e))
    assert not bad, bad


def _oracle_in_chunks(sd, x, nodes, graphs, img_h, chunk=4, dtype=None, want_feat=False):
    """The CPU oracle over `graphs` independent graphs, `chunk` at a time (bounds the host memory); dtype=torch.float64 runs
    the SAME restatement in double precision (the "exact" answer for the fp32 noise-floor tests below)."""
    from oracle import posenet_ref as O
    if dtype is not None:
        sd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
    oa, orr, of = [], [], []
    for g0 in range(0, graphs, chunk):
        g1 = min(graphs, g0 + chunk)
        st = {} if want_feat else None
        xx = x[g0 * nodes:g1 * nodes]
        a, r, _ = O.posenet_forward(sd, xx if dtype is None else xx.to(dtype), O.batch_edge_index(nodes, g1 - g0), img_h, 2, st)
        oa.append(a)
        orr.append(r)
        if want_feat:
            of.append(st["fc"])
    return torch.cat(oa), torch.cat(orr), (torch.cat(of) if want_feat else None)


@pytest.mark.parametrize("gnn_dtype", ["f32", "bf16"])
def test_configs2_bf16_forward_as_benched_vs_oracle(dev, gnn_dtype):
    """BASELINE.json configs[2] exactly as `bench.py --graphs 64 --encoder-dtype bf16 [--gnn-dtype bf16]` runs it: 64 graphs x 8
    nodes x 224x224 (512 images), two streams of 256 images -- the 512-image buffers, the two-stream split, the layer-1
    dispatch and the fused bf16 stem at FULL size, which the 2-graph bf16 tests never launch.  Against the
    fp32 oracle on all 64 graphs under the stated bf16 bars, per-graph worst case included."""
    m, sd = _r3_model(dev)
    G, N = 64, 8
    x = torch.randn((G * N, 3 * 224 * 224), generator=torch.Generator().manual_seed(2468))
    from relpose_gnn_amd.graph import fc_batch
    data = fc_batch(x, N).to(dev)
    m.hip_streams = 2
    m.encoder_dtype, m.gnn_dtype = "bf16", gnn_dtype
    try:
        a, r, _ = m(data)
        m.check_edge_index()
        feat = torch.cat([m._enc.run(m.feature_extractor.state_dict, "", data.x[i:i + 256].view(256, 3, 224, 224))
                          for i in (0, 256)]).cpu()
    finally:
        m.encoder_dtype, m.gnn_dtype = "f32", "f32"
    oa, orr, of = _oracle_in_chunks(sd, x, N, G, 224, want_feat=True)
    ef, ea, er = rel_err(feat, of), rel_err(a.cpu(), oa), rel_err(r.cpu(), orr)
    pg_r = max(rel_err(r[g * 56:(g + 1) * 56].cpu(), orr[g * 56:(g + 1) * 56]) for g in range(G))
    pg_f = max(rel_err(feat[g * 8:(g + 1) * 8], of[g * 8:(g + 1) * 8]) for g in range(G))
    _report({"case": f"configs2_64graphs_224px_2streams_bf16_encoder_{gnn_dtype}_gnn_vs_fp32_oracle", "feat_rel_err": ef,
             "abs_pose_rel_err": ea, "rel_pose_rel_err": er, "worst_graph_rel_pose_rel_err": pg_r,
             "worst_graph_feat_rel_err": pg_f})
    # Size-independent statement of the same bars: relative L2 error over the whole tensor.
    l2 = lambda got, ref: float((got.double() - ref.double()).norm() / ref.double().norm())
    l2f, l2a, l2r = l2(feat, of), l2(a.cpu(), oa), l2(r.cpu(), orr)
    _report({"case": f"configs2_64graphs_{gnn_dtype}_gnn_relative_l2", "feat": l2f, "abs_pose": l2a, "rel_pose": l2r})
    k = 1.0 if gnn_dtype == "f32" else 1.5            # ~20 more chained bf16-input GEMMs (same factor as the 2-graph test)
    # The max-norm bars BF16_REL / BF16_ABS were stated on 2-graph batches (672 / 96 pose components).  The maximum of N
    # roughly Gaussian errors grows like sqrt(2 ln N); here N = 64 x 336 / 64 x 48 components -> x 1.24 / x 1.37 (measured on
    # MI355X, r3: rel 2.06e-2 / 2.97e-2, abs 3.3e-2 / 6.9e-2 for the fp32 / bf16 GNN).  The L2 statement needs no such factor.
    import math
    ev_r = math.sqrt(math.log(G * 336) / math.log(2 * 336))
    ev_a = math.sqrt(math.log(G * 48) / math.log(2 * 48))
    assert ef < BF16_FEAT and er < k * ev_r * BF16_REL and ea < k * ev_a * BF16_ABS, (ef, ea, er)
    assert l2f < BF16_FEAT and l2r < k * BF16_REL and l2a < k * BF16_ABS, (l2f, l2a, l2r)
    # per graph the norm in the denominator is that graph's own (smaller than the batch-wide one): twice the bar
    assert pg_f < 2 * BF16_FEAT and pg_r < 2 * k * BF16_REL, (pg_f, pg_r)


@pytest.mark.parametrize("shape", ["configs1_224", "eval_256x341"])
def test_fp32_error_is_the_fp32_noise_floor(dev, shape):
    """the fp32 forward sits 1.2e-5 (configs[1]) / 6.7e-5 (256x341, abs poses) from the CPU fp32 oracle
    against a 1e-4 bar -- is that kernel error or the conditioning of the randomly initialised network?  Three seeds per
    shape; the float64 run of the same oracle is the exact answer.  Asserted: next to the 1e-4 bar vs the fp32 oracle, the
    HIP result is no further from the EXACT answer than 1.5 x the CPU fp32 reference itself is (the reference's own fp32
    rounding noise, amplified ~100x by the random abs-pose head, is the floor; a kernel bug would break the ratio).
    Round 5 (two-level accumulation in the fp32 Linears, RPG_TUNE_FOLD_K): the bar was 2.0 through round 4; measured ratios
    now 0.75 .. 1.21 on the abs poses and 0.84 .. 1.61 on the rel poses, the 1.61 being 3.2e-6 against 2.0e-6 -- two-graph
    maxima at a few ulp of the output, hence the floor of 3e-6 under the ratio (the fixed-pair test at 64 graphs x 256x341,
    tests/test_hip_eval_geometry.py, asserts 1.2 on numbers ten times larger)."""
    h, w, G = (224, 224, 2) if shape == "configs1_224" else (256, 341, 2)
    m, sd = _r3_model(dev, img_h=h)
    from relpose_gnn_amd.graph import fc_batch
    worst = {}
    for seed in (11, 12, 13):
        x = torch.randn((G * 8, 3 * h * w), generator=torch.Generator().manual_seed(seed))
        a, r, _ = m(fc_batch(x, 8).to(dev))
        oa, orr, _ = _oracle_in_chunks(sd, x, 8, G, h, chunk=2)
        oa64, or64, _ = _oracle_in_chunks(sd, x, 8, G, h, chunk=1, dtype=torch.float64)
        rec = {"case": f"fp32_noise_floor_{shape}_seed{seed}",
               "hip_vs_fp32_oracle_abs": rel_err(a.cpu(), oa), "hip_vs_fp32_oracle_rel": rel_err(r.cpu(), orr),
               "hip_vs_fp64_abs": rel_err(a.cpu(), oa64), "hip_vs_fp64_rel": rel_err(r.cpu(), or64),
               "cpu_fp32_vs_fp64_abs": rel_err(oa, oa64), "cpu_fp32_vs_fp64_rel": rel_err(orr, or64)}
        _report(rec)
        assert rec["hip_vs_fp32_oracle_abs"] < 1e-4 and rec["hip_vs_fp32_oracle_rel"] < 1e-4, rec
        for k in ("abs", "rel"):
            # floor of 3e-6: where the CPU reference happens to land within rounding of the exact answer the ratio is noise
            assert rec[f"hip_vs_fp64_{k}"] <= 1.5 * max(rec[f"cpu_fp32_vs_fp64_{k}"], 3e-6), rec
        worst[seed] = rec
