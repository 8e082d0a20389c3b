"""Host-side logic that needs no GPU: state-dict contract, weight packing, graph containers, and that the C-ABI
library loads and exports every symbol include/relpose_gnn_hip.h declares (no compute calls here)."""
import os
import re

import pytest
import torch

import relpose_gnn_amd.synth as S
from conftest import ROOT
from relpose_gnn_amd.graph import Batch, Data, fc_batch, fc_edge_index


def test_library_exports_every_declared_symbol():
    from relpose_gnn_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "relpose_gnn_hip.h")).read()
    declared = set(re.findall(r"\b(rpg_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = _lib.lib()                                    # loads without a GPU
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.rpg_abi_version() == 1
    # pure host-side argument validation paths (no kernel is launched)
    assert lib.rpg_maxpool3x3s2_nhwc_f32(None, None, 1, 4, 4, 8, None) == _lib.RPG_ERR_BAD_ARG
    assert lib.rpg_gnn_workspace_bytes(8, 56, 2048) > 5 * 56 * 2048 * 4
    assert lib.rpg_gnn_workspace_bytes(8, 56, 100) == 0          # d % 32 != 0
    planes = _lib.int_array([64, 128, 256, 512])
    ws = lib.rpg_resnet_workspace_bytes(256, 224, 224, planes)
    assert ws >= 4 * (256 * 224 * 224 * 4 + 256 * 112 * 112 * 64 + 4 * 256 * 56 * 56 * 64)
    with pytest.raises(ValueError):
        _lib.check(_lib.RPG_ERR_BAD_ARG, "x")


def test_tuning_and_timer_constants_match_the_header():
    """ops.TUNE_* and the timer classes are the header's #defines (they are passed across the ABI as plain ints), and
    rpg_set_tuning validates keys / values on the host side (no GPU involved)."""
    from relpose_gnn_amd import _lib, ops
    hdr = open(os.path.join(ROOT, "include", "relpose_gnn_hip.h")).read()
    defs = {k: int(v) for k, v in re.findall(r"#define\s+(RPG_[A-Z0-9_]+)\s+(-?\d+)\b", hdr)}
    tune = {k[len("RPG_TUNE_"):]: v for k, v in defs.items() if k.startswith("RPG_TUNE_")}
    assert len(set(tune.values())) == len(tune)                       # no two knobs share a key
    for name, key in tune.items():
        assert getattr(ops, "TUNE_" + name) == key, name
    assert {k for k in dir(ops) if k.startswith("TUNE_")} == {"TUNE_" + k for k in tune}
    lib = _lib.lib()
    assert lib.rpg_set_tuning(12345, 0) == _lib.RPG_ERR_BAD_ARG       # unknown key
    assert lib.rpg_set_tuning(defs["RPG_TUNE_BK"], 24) == _lib.RPG_ERR_BAD_ARG
    assert lib.rpg_set_tuning(defs["RPG_TUNE_WINOGRAD"], 7) == _lib.RPG_ERR_BAD_ARG
    for name, key in tune.items():                                    # defaults are accepted and restore the defaults
        default = {"TILE": -1, "BK": 0, "BF16_BK": 32, "WINO_SHORT": 0, "BF16_TILE": -1, "BF16_WS64": 0, "SK_MIN_ITS": 8, "INKERNEL_FIXUP": 0, "BF16_CHUNK": 0, "WINO2D": 0, "BF16_LINEAR_DMA": 0, "BF16_PERSIST": 0, "FOLD_K": 256, "LIN112": 3}.get(name, 1)
        assert lib.rpg_set_tuning(key, default) == 0, name
    # the weights-stationary probe kernel is not part of the product library (tools/probes/conv3x3_bf16_ws64.inc)
    assert lib.rpg_set_tuning(defs["RPG_TUNE_BF16_WS64"], 1) == _lib.RPG_ERR_BAD_ARG
    # ... and neither is the nested 2-D Winograd kernel (tools/probes/winograd2d.hip)
    assert lib.rpg_set_tuning(defs["RPG_TUNE_WINO2D"], 1) == _lib.RPG_ERR_BAD_ARG and lib.rpg_wino43_weights_floats(2, 3) == 18 * 6


def test_state_dict_contract_matches_reference_inventory():
    from relpose_gnn_amd.posenet import PoseNetX_R2
    from relpose_gnn_amd.resnet import resnet34
    m = PoseNetX_R2(resnet34(), droprate=0.0, pretrained=True, feat_dim=2048, edge_feat_dim=2048, node_dim=2048,
                    use_gnn=True, knn=-1, gnn_recursion=2, input_img_height=224)
    sd = m.state_dict()
    shapes = S.posenet_r2_param_shapes()                 # checked against the instantiated reference in make_golden.py
    assert list(sd.keys()) == list(shapes.keys()) and len(sd) == 248
    assert all(tuple(sd[k].shape) == tuple(v) for k, v in shapes.items())
    assert sum(p.numel() for p in m.parameters()) == 74805836
    assert m.feature_extractor.fc.in_features == 512 and m.feature_extractor.fc.out_features == 2048
    # pretrained=True re-initialises only the six top-level Linears (bias 0), posenet.py:981-997
    assert float(m.proj_edge.bias.abs().max()) == 0.0 and float(m.gnn1.mlp[0].bias.abs().max()) > 0.0
    m.load_state_dict(S.synth_state_dict(shapes, seed=3))
    with pytest.raises(RuntimeError):                    # no CPU fallback
        m(fc_batch(torch.zeros(8, 3 * 224 * 224), 8))
    with pytest.raises(NotImplementedError):
        PoseNetX_R2(resnet34(), use_gnn=False)
    # unequal dims: the reference's own forward raises a shape error for them (posenet.py:974-975,1085-1086)
    for dims in ((64, 32, 64), (64, 64, 32), (32, 64, 64)):
        with pytest.raises(ValueError):
            PoseNetX_R2(resnet34(), feat_dim=dims[0], edge_feat_dim=dims[1], node_dim=dims[2], use_gnn=True)
    # the other constructor flags keep the reference's state-dict inventory (checked against it in make_golden.py)
    m2 = PoseNetX_R2(resnet34(), feat_dim=64, edge_feat_dim=64, node_dim=64, use_gnn=True, use_attention=True,
                     use_AP=False, L=2)
    assert list(m2.state_dict().keys()) == list(S.posenet_r2_param_shapes(64, 64, 64, use_attention=True, use_AP=False, L=2).keys())
    assert m2.fc_xyz.in_features == 128
    assert torch.equal(m.compute_RP(torch.arange(12.).view(4, 3), torch.tensor([[0, 3], [1, 1]])),
                       torch.tensor([[-3., -3., -3.], [6., 6., 6.]]))


def test_weight_packing():
    from oracle import posenet_ref as O
    from relpose_gnn_amd.params import pack_gnn, pack_resnet
    planes, blocks = (8, 16, 32, 64), (1, 2, 1, 1)
    sd = S.synth_state_dict(S.posenet_r2_param_shapes(64, 64, 64, planes, blocks), seed=2)
    t, b, p = pack_resnet(sd)
    assert b == list(blocks) and p == list(planes)
    assert len(t) == 4 + 8 * 5 + 4 * 3 + 2                 # {w, scale, shift, u_wino} per conv; u is None without a GPU
    assert all(v is None for v in t[3:-2:4]) and all(v is not None for i, v in enumerate(t) if i % 4 != 3 or i >= len(t) - 2)
    assert t[0].shape == (8, 7, 7, 4) and float(t[0][..., 3].abs().max()) == 0.0
    assert torch.equal(t[0][..., :3], sd["feature_extractor.conv1.weight"].permute(0, 2, 3, 1))
    # folded BN == eval-mode batch norm
    x = S.hash_normal("pk.x", (2, 8, 5, 5))
    ref = O._bn(sd, "feature_extractor.bn1.", x)
    assert torch.allclose(x * t[1].view(1, -1, 1, 1) + t[2].view(1, -1, 1, 1), ref, atol=1e-6)
    g = pack_gnn(sd)
    assert len(g) == 26 and g[22].shape == (128, 64) and g[23].shape == (192, 64) and g[24].shape == (64, 64)
    assert torch.equal(g[23][128:], sd["gnn1.mlp.0.weight"][:, :64]) and torch.equal(g[25], sd["gnn1.mlp.0.weight"][:, 64:])
    assert g[10].shape == (24, 64) and g[18].shape == (6, 64) and g[21].shape == (6,)
    assert torch.equal(g[10][8:16], sd["gnn1.att.theta.weight"]) and torch.equal(g[20][3:], sd["fc_wpqr_R.weight"])
    bad = dict(sd)
    bad.pop("feature_extractor.layer2.0.downsample.0.weight")
    with pytest.raises(ValueError):
        pack_resnet(bad)


def test_graph_containers():
    from oracle import posenet_ref as O
    assert torch.equal(fc_edge_index(8), O.fc_edge_index(8)) and torch.equal(fc_edge_index(4), O.fc_edge_index(4))
    x = torch.arange(24.).view(12, 2)
    y = torch.arange(72.).view(12, 6)
    b = fc_batch(x, 4, y)
    assert torch.equal(b.edge_index, O.batch_edge_index(4, 3)) and b.num_graphs == 3 and len(b) == 3
    assert torch.equal(b.batch, torch.tensor([0] * 4 + [1] * 4 + [2] * 4))
    assert torch.equal(b.edge_attr, y[b.edge_index[1]] - y[b.edge_index[0]])          # dataset_7Scenes_multi.py:425-429
    graphs = [Data(x=x[i * 4:(i + 1) * 4], edge_index=fc_edge_index(4), y=y[i * 4:(i + 1) * 4]) for i in range(3)]
    c = Batch.from_data_list(graphs)
    assert torch.equal(c.edge_index, b.edge_index) and torch.equal(c.x, x) and torch.equal(c.batch, b.batch)
    d = c.to("cpu")
    assert d is not c and torch.equal(d.x, c.x)


def test_synth_is_deterministic_and_reasonable():
    a = S.hash_normal("w", (1000,), 2.0, 1.0, seed=5)
    b = S.hash_normal("w", (1000,), 2.0, 1.0, seed=5)
    assert torch.equal(a, b) and not torch.equal(a, S.hash_normal("w", (1000,), 2.0, 1.0, seed=6))
    big = S.hash_normal("n", (200000,))
    assert abs(float(big.mean())) < 0.01 and abs(float(big.std()) - 1.0) < 0.01
    u = S.hash_range("u", (1000,), 0.5, 1.5)
    assert float(u.min()) >= 0.5 and float(u.max()) <= 1.5
    # pinned values: the GPU box must regenerate exactly these (the goldens were produced from this generator)
    assert [round(v, 5) for v in S.hash_normal("pin", (3,)).tolist()] == [-1.97496, -0.28621, -1.54145]


def test_stream_partition_from_pyg_batch_tables():
    """A torch_geometric Batch (what the reference's DataLoader hands the model, testing/test.py:193) carries no
    `graph_sizes`; the cut at graph boundaries comes from its collation tables or from ptr / batch."""
    import types
    from relpose_gnn_amd.graph import fc_batch
    from relpose_gnn_amd.posenet import PoseNetX_R2
    from relpose_gnn_amd.resnet import ResNet
    ref = fc_batch(torch.zeros(8 * 6, 12), 8)
    n, e = 48, 56 * 6
    want = ([8] * 6, [56] * 6)
    cum_n, cum_e = torch.arange(7) * 8, torch.arange(7) * 56
    pyg2 = types.SimpleNamespace(x=ref.x, edge_index=ref.edge_index, batch=ref.batch,
                                 _slice_dict={"x": cum_n, "edge_index": cum_e, "y": cum_n}, num_graphs=6)
    pyg1 = types.SimpleNamespace(x=ref.x, edge_index=ref.edge_index, batch=ref.batch,
                                 __slices__={"x": cum_n.tolist(), "edge_index": cum_e.tolist()})
    with_ptr = types.SimpleNamespace(x=ref.x, edge_index=ref.edge_index, batch=ref.batch, ptr=cum_n)
    only_batch = types.SimpleNamespace(x=ref.x, edge_index=ref.edge_index, batch=ref.batch)
    for d in (ref, pyg2, pyg1, with_ptr, only_batch):
        assert PoseNetX_R2._graph_sizes(d, n, e) == want
    # ragged graphs through `batch` only
    from relpose_gnn_amd.graph import Batch, Data, fc_edge_index
    rag = Batch.from_data_list([Data(x=torch.zeros(k, 12), edge_index=fc_edge_index(k)) for k in (3, 8, 5, 2)])
    bare = types.SimpleNamespace(x=rag.x, edge_index=rag.edge_index, batch=rag.batch)
    assert PoseNetX_R2._graph_sizes(bare, 18, rag.edge_index.shape[1]) == ([3, 8, 5, 2], [6, 56, 20, 2])
    # edges not grouped by graph: no contiguous cut
    perm = torch.randperm(rag.edge_index.shape[1], generator=torch.Generator().manual_seed(0))
    mixed = types.SimpleNamespace(x=rag.x, edge_index=rag.edge_index[:, perm], batch=rag.batch)
    assert PoseNetX_R2._graph_sizes(mixed, 18, rag.edge_index.shape[1]) is None
    # stale tables are rejected
    assert PoseNetX_R2._graph_sizes(pyg2, n + 8, e) is None
    m = PoseNetX_R2(ResNet((1, 1, 1, 1), (8, 16, 32, 64)), droprate=0.0, pretrained=False, feat_dim=64, edge_feat_dim=64,
                    node_dim=64, use_gnn=True)
    assert m._partition(pyg2, n, e) == [(0, 24, 0, 168, 0), (24, 48, 168, 336, 1)]
    assert m._partition(types.SimpleNamespace(x=ref.x[:8], edge_index=ref.edge_index[:, :56], batch=ref.batch[:8],
                                              num_graphs=1), 8, 56) is None
    # explicit schedules must cover every graph exactly once (equal total length is not enough)
    m.stream_schedule = [(0, 2, 0), (2, 6, 1)]
    assert m._partition(pyg2, n, e) == [(0, 16, 0, 112, 0), (16, 48, 112, 336, 1)]
    m.stream_schedule = [(3, 6, 1), (0, 3, 0)]                     # issue order is the caller's business
    assert m._partition(pyg2, n, e) == [(24, 48, 168, 336, 1), (0, 24, 0, 168, 0)]
    for bad in ([(0, 3, 0), (2, 5, 1)], [(0, 2, 0), (1, 3, 1), (3, 6, 0)], [(0, 3, 0), (4, 6, 1)], [(0, 3, 0), (3, 5, 1)],
                [(0, 4, 0), (2, 4, 1), (4, 6, 1)]):
        m.stream_schedule = bad
        with pytest.raises(ValueError):
            m._partition(pyg2, n, e)
    m.stream_schedule = None


def test_stem_pair_table_matches_the_kernel():
    """params.stem_pair_table (what pack_stem_pairs packs by) == rpg_stem_pair_table (what csrc/stem.hip reads by)."""
    import ctypes as C
    from relpose_gnn_amd import _lib
    from relpose_gnn_amd.params import pack_stem_pairs, stem_pair_table
    ta, tb = (C.c_int * 222)(), (C.c_int * 222)()
    assert _lib.lib().rpg_stem_pair_table(ta, tb) == 0
    a, b = stem_pair_table()
    assert len(a) == len(b) == 74
    assert [tuple(ta[3 * i: 3 * i + 3]) for i in range(74)] == a
    assert [None if tb[3 * i] < 0 else tuple(tb[3 * i: 3 * i + 3]) for i in range(74)] == b
    taps = sorted(a + [t for t in b if t is not None])
    assert taps == sorted((c, kh, kw) for c in range(3) for kh in range(7) for kw in range(7))      # every tap exactly once
    w = torch.arange(64 * 147, dtype=torch.float32).view(64, 3, 7, 7)
    sc = torch.full((64,), 2.0)
    wp_all = pack_stem_pairs(w, sc)
    assert wp_all.shape == ((74 + 75) * 2 * 64,)                                                    # tile-kernel image, then the strip-march image
    wp = wp_all[:74 * 2 * 64].view(74, 2, 64)
    st = wp_all[74 * 2 * 64:].view(2, 75, 64)                                                       # [nf][MFMA of a convolution row][lane]
    # strip image: kernel row 0, channel 1, horizontal pair 2 = entry 3 * 1 + 2: taps (1, 0, 4) | (1, 0, 5); entry 9 * 1 + 9 + 2 = the
    # vertical pair (kh 0 | 1, kw 6) of channel 2; the last three entries are the lone (6, 6) taps with a zero partner
    assert float(st[1, 5, 7]) == 2.0 * float(w[39, 1, 0, 4]) and float(st[1, 5, 32 + 7]) == 2.0 * float(w[39, 1, 0, 5])
    assert float(st[0, 20, 3]) == 2.0 * float(w[3, 2, 0, 6]) and float(st[0, 20, 32 + 3]) == 2.0 * float(w[3, 2, 1, 6])
    assert float(st[0, 74, 3]) == 2.0 * float(w[3, 2, 6, 6]) and float(st[0, 74, 32 + 3]) == 0.0
    assert float(wp[5, 1, 7]) == 2.0 * float(w[39, a[5][0], a[5][1], a[5][2]])                      # first tap, channel 32 + 7
    assert float(wp[5, 1, 32 + 7]) == 2.0 * float(w[39, b[5][0], b[5][1], b[5][2]])                # second tap
    assert float(wp[73, 0, 40]) == 0.0                                                              # the missing partner


def test_host_f32_to_bf16_is_round_to_nearest_even():
    """rpg_host_f32_to_bf16 (the staging threads of evaluate_stream round the bf16 encoder's node images with it) against torch's
    fp32 -> bf16 conversion -- which is also what the stem kernel's own (__bf16)float does -- on normal values across the whole
    exponent range, ties, zeros, infinities, denormals, the largest finite value (rounds to inf) and every NaN payload class."""
    from relpose_gnn_amd import _lib
    lib = _lib.lib()
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1 << 18, generator=g) * torch.logspace(-40, 38, 1 << 18)
    x[:10] = torch.tensor([float("nan"), float("inf"), -float("inf"), 0.0, -0.0, 1e-45, 3.3895313892515355e38, -1.0, 1.00390625, 1.01171875])
    bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (1 << 18,), dtype=torch.int64, generator=g).to(torch.int32).view(torch.float32)
    for t in (x, bits):
        out = torch.empty(t.shape, dtype=torch.bfloat16)
        assert lib.rpg_host_f32_to_bf16(t.data_ptr(), out.data_ptr(), t.numel()) == _lib.RPG_OK
        ref = t.bfloat16()
        nan = torch.isnan(ref)
        assert torch.equal(out.view(torch.int16)[~nan], ref.view(torch.int16)[~nan])
        assert torch.isnan(out[nan]).all() and nan.sum() == torch.isnan(t).sum()
    assert lib.rpg_host_f32_to_bf16(None, None, 0) == _lib.RPG_OK and lib.rpg_host_f32_to_bf16(None, None, 4) == _lib.RPG_ERR_BAD_ARG
    # round 6: the vector variants (AVX2 / AVX-512F where this CPU has them) are the scalar loop's arithmetic on 8 / 16 lanes: bit for bit
    # the same, at every length (scalar head up to the 32-byte store alignment, vector body, scalar tail) and destination alignment
    both = torch.cat([x, bits])
    want = torch.empty(both.shape, dtype=torch.bfloat16)
    assert lib.rpg_host_f32_to_bf16_isa(both.data_ptr(), want.data_ptr(), both.numel(), 1) == _lib.RPG_OK
    ran = 0
    for isa in (0, 2, 3):
        for n, off in ((both.numel(), 0), (1000, 3), (31, 1), (16, 0), (15, 5), (1, 0), (0, 0)):
            out = torch.full((n + off + 16,), 7.0, dtype=torch.bfloat16)
            rc = lib.rpg_host_f32_to_bf16_isa(both.data_ptr(), out.data_ptr() + 2 * off, n, isa)
            if rc == _lib.RPG_ERR_BAD_ARG and isa in (2, 3):
                break                                                              # this CPU lacks the instruction set
            assert rc == _lib.RPG_OK
            assert torch.equal(out.view(torch.int16)[off:off + n], want.view(torch.int16)[:n]), (isa, n, off)
            assert bool((out[:off] == 7.0).all()) and bool((out[off + n:] == 7.0).all()), (isa, n, off)      # nothing written outside
            ran += 1
    assert ran >= 7
    assert lib.rpg_host_f32_to_bf16_isa(both.data_ptr(), want.data_ptr(), 4, 9) == _lib.RPG_ERR_BAD_ARG
