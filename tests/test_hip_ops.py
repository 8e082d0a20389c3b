"""GPU parity tests, one per C-ABI entry point: HIP kernel vs a plain PyTorch-CPU fp32 statement of the same op
(the pieces of oracle/posenet_ref.py).  Tolerance 1e-5 relative (max-norm) per op -- an order below the 1e-4 the
north-star allows for the whole forward; index work (graph_prepare, edge gather, max-pool) is bit-exact."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def test_nchw_to_nhwc4(dev):
    from relpose_gnn_amd import ops
    x = _rand(3, 3, 17, 23)
    y = ops.nchw3_to_nhwc4(x.to(dev)).cpu()
    ref = F.pad(x.permute(0, 2, 3, 1), (0, 1))
    assert torch.equal(y, ref)


@pytest.mark.parametrize("n,h,w,cin,cout,k,stride,pad,res,relu", [
    (2, 9, 11, 8, 16, 3, 1, 1, False, True),       # ragged M (198 rows), tiny channels, 64x64 tile
    (1, 16, 16, 4, 64, 7, 2, 3, False, True),      # stem shape: Cin padded to 4, K=196 (tail masked)
    (2, 14, 14, 64, 128, 3, 2, 1, False, True),    # strided 3x3
    (2, 14, 14, 64, 128, 1, 2, 0, False, False),   # downsample 1x1/2, no activation
    (2, 28, 28, 128, 128, 3, 1, 1, True, True),    # residual + relu
    (40, 56, 56, 64, 64, 3, 1, 1, True, True),     # layer1 shape, M=125440 -> 256x64 tile path (N<=64, M>=65536)
    (16, 28, 28, 128, 256, 3, 1, 1, False, True),  # M=12544, N=256 -> tiles: 98*2=196 <384 -> 64x64
    (64, 28, 28, 128, 128, 3, 1, 1, True, True),   # M=50176, N=128 -> 392 tiles of 128x128
])
def test_conv_bn_act(dev, n, h, w, cin, cout, k, stride, pad, res, relu):
    from relpose_gnn_amd import ops
    x = _rand(n, cin, h, w, seed=1)
    wt = _rand(cout, cin, k, k, seed=2, scale=(2.0 / (cin * k * k)) ** 0.5)
    scale = torch.rand(cout, generator=torch.Generator().manual_seed(3)) + 0.5
    shift = _rand(cout, seed=4, scale=0.1)
    ref = F.conv2d(x, wt, None, stride=stride, padding=pad) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    r = None
    if res:
        r = _rand(*ref.shape, seed=5)
        ref = ref + r
    if relu:
        ref = F.relu(ref)
    y = ops.conv2d_bn_act_nhwc(x.permute(0, 2, 3, 1).contiguous().to(dev), wt.permute(0, 2, 3, 1).contiguous().to(dev),
                               scale.to(dev), shift.to(dev), None if r is None else r.permute(0, 2, 3, 1).contiguous().to(dev),
                               stride=stride, pad=pad, relu=relu)
    assert y.shape == (n, ref.shape[2], ref.shape[3], cout)
    err = rel_err(y.cpu().permute(0, 3, 1, 2), ref)
    assert err < TOL, err


def test_conv_transpose_detecting(dev):
    """A = identity-like input, asymmetric weights: catches a row/col swap in the MFMA C layout."""
    from relpose_gnn_amd import ops
    cin, cout = 64, 128
    x = torch.zeros(1, cin, 8, 8)
    for c in range(cin):
        x[0, c, c % 8, c // 8] = 1.0 + c
    wt = torch.arange(cout * cin, dtype=torch.float32).view(cout, cin, 1, 1) * 1e-3
    ref = F.conv2d(x, wt)
    y = ops.conv2d_bn_act_nhwc(x.permute(0, 2, 3, 1).contiguous().to(dev), wt.permute(0, 2, 3, 1).contiguous().to(dev), None, None)
    assert rel_err(y.cpu().permute(0, 3, 1, 2), ref) < 1e-6


@pytest.mark.parametrize("kernel", [1, 1 + 128, 1 + 128 + (7 << 8), 1 + 128 + (3 << 8)])      # RPG_TUNE_FUSED_STEM: tile kernel (default), strip-march kernel (bit 7: default bands / 7- / 3-row bands)
@pytest.mark.parametrize("n,h,w", [(2, 224, 224), (1, 256, 341), (3, 64, 64), (2, 37, 53), (1, 7, 9), (1, 450, 600), (5, 32, 40), (1, 1, 1), (9, 250, 123),
                                   (2, 30, 130)])
def test_fused_stem_conv_bn_relu_maxpool(dev, n, h, w, kernel):
    """rpg_stem_conv7x7s2_bn_relu_maxpool_f32 vs conv2d(7x7, s2, p3) -> BN affine -> ReLU -> max_pool2d(3, 2, 1) (the
    torchvision stem reached from posenet.py:1037): 224x224 (one column tile of 56), the 256x341 evaluation shape (two
    column tiles), odd / tiny sizes (ragged tiles, all-border tiles) and a wide image (three column tiles).
    Round 6: also the strip-march kernel (opt-in, bit 7; csrc/stem.hip stem_strip_f32_kernel: strips of 15 pooled columns, bands of
    pooled rows, odd convolution heights and widths, images narrower than a strip)."""
    from relpose_gnn_amd import ops
    from relpose_gnn_amd.params import pack_stem_pairs
    ops.set_tuning(ops.TUNE_FUSED_STEM, kernel)
    try:
        _fused_stem_f32_case(dev, n, h, w)
    finally:
        ops.set_tuning(ops.TUNE_FUSED_STEM, 1)


def _fused_stem_f32_case(dev, n, h, w):
    from relpose_gnn_amd import ops
    from relpose_gnn_amd.params import pack_stem_pairs
    x = _rand(n, 3, h, w, seed=h)
    wt = _rand(64, 3, 7, 7, seed=2, scale=(2.0 / 147) ** 0.5)
    g = torch.Generator().manual_seed(3)
    scale = (torch.rand(64, generator=g) + 0.5) * torch.where(torch.rand(64, generator=g) < 0.15, -1.0, 1.0)   # some negative gammas
    shift = _rand(64, seed=4, scale=0.3)
    ref = F.max_pool2d(F.relu(F.conv2d(x, wt, None, stride=2, padding=3) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)), 3, 2, 1)
    y = ops.stem_conv_bn_relu_maxpool(x.to(dev), pack_stem_pairs(wt, scale).to(dev), shift.to(dev))
    assert y.shape == (n, ref.shape[2], ref.shape[3], 64)
    assert rel_err(y.cpu().permute(0, 3, 1, 2), ref) < 1e-5


def test_maxpool_and_avgpool(dev):
    from relpose_gnn_amd import ops
    x = _rand(3, 16, 13, 18, seed=7)
    xn = x.permute(0, 2, 3, 1).contiguous().to(dev)
    assert torch.equal(ops.maxpool3x3s2_nhwc(xn).cpu().permute(0, 3, 1, 2), F.max_pool2d(x, 3, 2, 1))
    assert rel_err(ops.global_avgpool_nhwc(xn).cpu(), x.mean(dim=(2, 3))) < 1e-6


def _fc_edges(n, b):
    from relpose_gnn_amd.graph import fc_edge_index
    ei = fc_edge_index(n)
    return torch.cat([ei + g * n for g in range(b)], 1)


@pytest.mark.parametrize("n_nodes,b", [(8, 1), (8, 5), (4, 3), (8, 300)])
def test_graph_prepare_fc(dev, n_nodes, b):
    from relpose_gnn_amd import ops
    ei = _fc_edges(n_nodes, b)
    n = n_nodes * b
    g = ops.graph_prepare(ei.to(dev), n)
    assert int(g["status"].item()) == 0
    ends = g["ends"].cpu()
    assert torch.equal(ends[0], ei[0]) and torch.equal(ends[1], ei[1])
    assert torch.equal(ends[2], torch.minimum(ei[0], ei[1])) and torch.equal(ends[3], torch.maximum(ei[0], ei[1]))
    rowptr, perm = g["rowptr"].cpu().long(), g["perm"].cpu().long()
    order = torch.sort(ei[1], stable=True).indices          # edges grouped by target, ascending edge id inside
    assert torch.equal(perm, order)
    assert torch.equal(rowptr, torch.cat([torch.zeros(1, dtype=torch.long), torch.bincount(ei[1], minlength=n).cumsum(0)]))


def test_graph_prepare_ragged_and_invalid(dev):
    from relpose_gnn_amd import ops
    g0 = torch.Generator().manual_seed(11)
    n, e = 37, 501
    ei = torch.randint(0, n, (2, e), generator=g0)
    ei[1, :40] = 5                                  # a hub node and several isolated ones
    g = ops.graph_prepare(ei.to(dev), n)
    assert int(g["status"].item()) == 0
    assert torch.equal(g["perm"].cpu().long(), torch.sort(ei[1], stable=True).indices)
    bad = ei.clone()
    bad[0, 3], bad[1, 9] = n, -1
    g = ops.graph_prepare(bad.to(dev), n)
    assert int(g["status"].item()) == 2
    assert int(g["rowptr"].cpu()[-1]) == e - 2      # the two bad edges are not in the CSR
    assert int(g["ends"].max()) < n and int(g["ends"].min()) >= 0


def test_edge_concat_gather(dev):
    from relpose_gnn_amd import ops
    from oracle.posenet_ref import edge_concat
    x = _rand(24, 64, seed=3)
    ei = _fc_edges(8, 3)
    assert torch.equal(ops.edge_concat_gather(x.to(dev), ei.to(dev)).cpu(), edge_concat(x, ei))


@pytest.mark.parametrize("m,widths,n_out,gather,res,relu", [
    (56, (64, 64, 64), 64, True, False, True),        # edge_mlp.0 at D=64
    (8, (64,), 192, False, False, False),             # att g|theta|phi at D=64
    (56, (8,), 64, False, True, False),               # att.W: K=8 (< BK), residual
    (1792, (256, 256), 256, True, False, True),       # ragged-free mid size
    (300, (128, 64), 96, True, False, False),         # N not a multiple of 64, M ragged
    (2048, (512,), 2048, False, False, False),        # 128x128 tile path (16*16=256 tiles <384 -> 64x64) large K
    (6272, (512, 512, 512), 1024, True, True, True),  # 49*8=392 tiles -> 128x128 path, three gathered sources
])
def test_linear_gather(dev, m, widths, n_out, gather, res, relu):
    from relpose_gnn_amd import ops
    g0 = torch.Generator().manual_seed(5)
    rows = 97
    srcs, cpu_cat = [], []
    for i, wd in enumerate(widths):
        if gather:
            a = _rand(rows, wd, seed=10 + i)
            idx = torch.randint(0, rows, (m,), generator=g0)
            srcs.append((a.to(dev), idx.to(dev)))
            cpu_cat.append(a[idx])
        else:
            a = _rand(m, wd, seed=10 + i)
            srcs.append((a.to(dev), None))
            cpu_cat.append(a)
    k = sum(widths)
    wt = _rand(n_out, k, seed=20, scale=k ** -0.5)
    bias = _rand(n_out, seed=21, scale=0.1)
    ref = F.linear(torch.cat(cpu_cat, 1), wt, bias)
    r = None
    if res:
        r = _rand(m, n_out, seed=22)
        ref = ref + r
    if relu:
        ref = F.relu(ref)
    y = ops.linear_gather(srcs, wt.to(dev), bias.to(dev), m, None if r is None else r.to(dev), relu)
    err = rel_err(y.cpu(), ref)
    assert err < TOL, err


@pytest.mark.parametrize("r,c", [(56, 8), (7, 256), (130, 64)])
def test_attention_rows(dev, r, c):
    from relpose_gnn_amd import ops
    gtp = _rand(r, 3 * c, seed=9, scale=1.5)
    g, th, ph = gtp[:, :c], gtp[:, c:2 * c], gtp[:, 2 * c:]
    a = torch.softmax(ph.unsqueeze(2) * th.unsqueeze(1), dim=-1)
    ref = torch.bmm(a, g.unsqueeze(2)).squeeze(2)
    err = rel_err(ops.attention_rows(gtp.to(dev)).cpu(), ref)
    assert err < TOL, err


@pytest.mark.parametrize("n_nodes,b,d", [(8, 4, 64), (8, 32, 2048), (4, 1, 128)])
def test_scatter_mean_bit_exact(dev, n_nodes, b, d):
    from relpose_gnn_amd import ops
    from oracle.posenet_ref import scatter_mean
    ei = _fc_edges(n_nodes, b)
    n = n_nodes * b
    msg = _rand(ei.shape[1], d, seed=13)
    g = ops.graph_prepare(ei.to(dev), n)
    out = ops.scatter_mean(msg.to(dev), g["rowptr"], g["perm"], n).cpu()
    ref = scatter_mean(msg, ei[1], n)
    assert torch.equal(out, ref), rel_err(out, ref)


def test_scatter_mean_isolated_nodes(dev):
    from relpose_gnn_amd import ops
    from oracle.posenet_ref import scatter_mean
    ei = torch.tensor([[0, 1, 2, 2], [3, 3, 3, 0]])
    msg = _rand(4, 32, seed=1)
    g = ops.graph_prepare(ei.to(dev), 6)
    out = ops.scatter_mean(msg.to(dev), g["rowptr"], g["perm"], 6).cpu()
    assert torch.equal(out, scatter_mean(msg, ei[1], 6))
    assert float(out[[1, 2, 4, 5]].abs().max()) == 0.0


@pytest.mark.parametrize("sizes,c,d", [((8, 8, 8, 8), 256, 2048), ((8, 4, 1, 6), 8, 64), ((5, 3), 64, 200)])
def test_attention_aggregate(dev, sizes, c, d):
    """rpg_attention_aggregate_f32 (attention rows + mean aggregation in one kernel; att.py:20-31 + PyG aggr='mean',
    my_gnn_layer.py:279,301): mbar without bias is BIT-EXACT scatter-mean of the messages (oracle, ascending edge order,
    isolated nodes -> 0), ybar = scatter-mean of the attention rows (1e-5), the bias lands only on nodes with incoming
    edges, and W ybar + mbar equals the reference's mean of (W y + b + msg)."""
    from relpose_gnn_amd import ops
    from oracle.posenet_ref import fc_edge_index, scatter_mean
    eis, off = [], 0
    for n_nodes in sizes:
        if n_nodes > 1:
            eis.append(fc_edge_index(n_nodes) + off)
        off += n_nodes
    n = off + 2                                             # two trailing nodes without any edge
    ei = torch.cat(eis, 1)
    e = ei.shape[1]
    gtp = _rand(e, 3 * c, seed=9, scale=1.5)
    msg = _rand(e, d, seed=10)
    bias = _rand(d, seed=11)
    g_, th, ph = gtp[:, :c], gtp[:, c:2 * c], gtp[:, 2 * c:]
    y = torch.bmm(torch.softmax(ph.unsqueeze(2) * th.unsqueeze(1), dim=-1), g_.unsqueeze(2)).squeeze(2)
    gp = ops.graph_prepare(ei.to(dev), n)
    ybar, mbar = ops.attention_aggregate(gtp.to(dev), msg.to(dev), gp["rowptr"], gp["perm"], n)
    assert torch.equal(mbar.cpu(), scatter_mean(msg, ei[1], n))                       # bit-exact (A9)
    assert rel_err(ybar.cpu(), scatter_mean(y, ei[1], n)) < TOL
    assert float(ybar[-2:].abs().max()) == 0.0 and float(mbar[-2:].abs().max()) == 0.0
    _, mb = ops.attention_aggregate(gtp.to(dev), msg.to(dev), gp["rowptr"], gp["perm"], n, bias.to(dev))
    has_in = torch.zeros(n, dtype=torch.bool)
    has_in[ei[1]] = True
    assert rel_err(mb.cpu(), scatter_mean(msg, ei[1], n) + bias * has_in.unsqueeze(1)) < 1e-6
    assert float(mb[-2:].abs().max()) == 0.0
    # the restructured aggregate == the reference's order (per-edge att, then mean)
    w = _rand(d, c, seed=12, scale=c ** -0.5)
    ref = scatter_mean(F.linear(y, w, bias) + msg, ei[1], n)
    got = ops.linear_gather([(ybar, None)], w.to(dev), None, n, residual=mb)
    assert rel_err(got.cpu(), ref) < TOL


def test_pose_heads(dev):
    from relpose_gnn_amd import ops
    x, w6, b6 = _rand(61, 2048, seed=2), _rand(6, 2048, seed=3, scale=0.02), _rand(6, seed=4)
    err = rel_err(ops.pose_heads(x.to(dev), w6.to(dev), b6.to(dev)).cpu(), F.linear(x, w6, b6))
    assert err < TOL, err


def test_bad_arguments_raise(dev):
    from relpose_gnn_amd import ops
    with pytest.raises(RuntimeError):
        ops.maxpool3x3s2_nhwc(torch.zeros(1, 4, 4, 8))                       # CPU tensor: no fallback
    with pytest.raises(ValueError):
        ops.maxpool3x3s2_nhwc(torch.zeros(1, 4, 4, 6, device=dev))           # c % 4 != 0
    with pytest.raises(TypeError):
        ops.pose_heads(torch.zeros(2, 8, device=dev, dtype=torch.float64), torch.zeros(6, 8, device=dev), torch.zeros(6, device=dev))


@pytest.mark.parametrize("bk", [16, 32])
@pytest.mark.parametrize("epi", [0, 1])
@pytest.mark.parametrize("tile", [-1, 0, 1, 2, 3])
@pytest.mark.parametrize("streamk", [0, 1])
def test_tile_engine_variants(dev, bk, epi, tile, streamk):
    """Every (K-step, epilogue, workgroup-tile) variant of the MFMA tile engine on a conv with ragged M, K tail,
    residual + ReLU, and on a 3-source gathered Linear whose N is not a tile multiple."""
    from relpose_gnn_amd import ops
    try:
        ops.set_tuning(ops.TUNE_BK, bk)
        ops.set_tuning(ops.TUNE_EPILOGUE, epi)
        ops.set_tuning(ops.TUNE_TILE, tile)
        ops.set_tuning(ops.TUNE_STREAMK, streamk)
        n, h, w, cin, cout = 3, 13, 17, 24, 72            # M = 663, K = 216 (not a multiple of 16/32), N = 72
        x = _rand(n, cin, h, w, seed=1)
        wt = _rand(cout, cin, 3, 3, seed=2, scale=(2.0 / (cin * 9)) ** 0.5)
        scale = torch.rand(cout, generator=torch.Generator().manual_seed(3)) + 0.5
        shift = _rand(cout, seed=4, scale=0.1)
        r = _rand(n, cout, h, w, seed=5)
        ref = F.relu(F.conv2d(x, wt, None, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + r)
        y = ops.conv2d_bn_act_nhwc(x.permute(0, 2, 3, 1).contiguous().to(dev), wt.permute(0, 2, 3, 1).contiguous().to(dev),
                                   scale.to(dev), shift.to(dev), r.permute(0, 2, 3, 1).contiguous().to(dev), stride=1, pad=1,
                                   relu=True)
        assert rel_err(y.cpu().permute(0, 3, 1, 2), ref) < TOL
        g0 = torch.Generator().manual_seed(5)
        m, widths, n_out = 333, (40, 24, 536), 100         # K = 600: every tile is split over several stream-K segments
        srcs, cat = [], []
        for i, wd in enumerate(widths):
            a = _rand(50, wd, seed=10 + i)
            idx = torch.randint(0, 50, (m,), generator=g0)
            srcs.append((a.to(dev), idx.to(dev)))
            cat.append(a[idx])
        k = sum(widths)
        wl, bias, res = _rand(n_out, k, seed=20, scale=k ** -0.5), _rand(n_out, seed=21), _rand(m, n_out, seed=22)
        ref = F.linear(torch.cat(cat, 1), wl, bias) + res
        out = ops.linear_gather(srcs, wl.to(dev), bias.to(dev), m, res.to(dev), False)
        assert rel_err(out.cpu(), ref) < TOL
    finally:
        ops.set_tuning(ops.TUNE_BK, 0)
        ops.set_tuning(ops.TUNE_EPILOGUE, 1)
        ops.set_tuning(ops.TUNE_TILE, -1)
        ops.set_tuning(ops.TUNE_STREAMK, 1)


@pytest.mark.parametrize("tile", [0, 3])
@pytest.mark.parametrize("streamk", [0, 1])
def test_tile_engine_eight_wave_workgroups(dev, tile, streamk):
    """The 8-wave instantiation of the tile engine (two waves per SIMD; 128x128 and 128x64 tiles, K step 32, buffer loaders)
    against torch and against the 4-wave one (RPG_TUNE_WAVES8 = 0): a strided 3x3 convolution with residual whose tile count
    leaves a stream-K remainder, a 1x1/2 convolution, and the gathered Linears of the GNN (two gathered residual rows in the
    epilogue, K = 2048 and K = 256) at M = 1792 / 256 rows."""
    from relpose_gnn_amd import ops
    try:
        ops.set_tuning(ops.TUNE_TILE, tile)
        ops.set_tuning(ops.TUNE_STREAMK, streamk)
        for ci, (n, h, w, cin, cout, k, st, pad, res) in enumerate([(9, 28, 28, 128, 256, 3, 2, 1, True), (7, 30, 22, 64, 160, 1, 2, 0, False)]):
            x = _rand(n, cin, h, w, seed=140 + ci)
            wt = _rand(cout, cin, k, k, seed=150 + ci, scale=(2.0 / (cin * k * k)) ** 0.5)
            scale = torch.rand(cout, generator=torch.Generator().manual_seed(160 + ci)) + 0.5
            shift = _rand(cout, seed=170 + ci, scale=0.1)
            ref = F.conv2d(x, wt, None, stride=st, padding=pad) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
            r = _rand(*ref.shape, seed=180 + ci) if res else None
            ref = F.relu(ref + r if res else ref)
            args = (x.permute(0, 2, 3, 1).contiguous().to(dev), wt.permute(0, 2, 3, 1).contiguous().to(dev), scale.to(dev),
                    shift.to(dev), None if r is None else r.permute(0, 2, 3, 1).contiguous().to(dev))
            y8 = ops.conv2d_bn_act_nhwc(*args, stride=st, pad=pad, relu=True)
            ops.set_tuning(ops.TUNE_WAVES8, 0)
            y4 = ops.conv2d_bn_act_nhwc(*args, stride=st, pad=pad, relu=True)
            ops.set_tuning(ops.TUNE_WAVES8, 1)
            assert rel_err(y8.cpu().permute(0, 3, 1, 2), ref) < TOL and rel_err(y8, y4) < TOL, ci
        g0 = torch.Generator().manual_seed(9)
        for m, widths, n_out in [(1792, (2048,), 768), (1792, (256,), 2048), (256, (2048, 2048), 512), (300, (64, 32, 160), 132)]:
            srcs, cat = [], []
            for i, wd in enumerate(widths):
                a = _rand(m, wd, seed=110 + i)
                srcs.append((a.to(dev), None))
                cat.append(a)
            kk = sum(widths)
            wl, bias, res = _rand(n_out, kk, seed=120, scale=kk ** -0.5), _rand(n_out, seed=121), _rand(m, n_out, seed=122)
            ref = F.relu(F.linear(torch.cat(cat, 1), wl, bias) + res)
            out8 = ops.linear_gather(srcs, wl.to(dev), bias.to(dev), m, res.to(dev), True)
            ops.set_tuning(ops.TUNE_WAVES8, 0)
            out4 = ops.linear_gather(srcs, wl.to(dev), bias.to(dev), m, res.to(dev), True)
            ops.set_tuning(ops.TUNE_WAVES8, 1)
            assert rel_err(out8.cpu(), ref) < TOL and rel_err(out8, out4) < TOL, (m, widths, n_out)
    finally:
        ops.set_tuning(ops.TUNE_WAVES8, 1)
        ops.set_tuning(ops.TUNE_TILE, -1)
        ops.set_tuning(ops.TUNE_STREAMK, 1)


@pytest.mark.parametrize("bk", [16, 32])
@pytest.mark.parametrize("tile", [-1, 0, 1, 2, 3])
@pytest.mark.parametrize("streamk", [0, 1])
def test_tile_engine_buffer_loaders(dev, bk, tile, streamk):
    """The buffer-load loaders + interleaved main loop (ConvLoaderB, ConvLoaderTap, GatherLoaderB; used whenever every K
    segment is a multiple of the K step) in every tile variant: a strided padded 3x3 conv (Cin = 64), the 7x7/2
    4-channel stem (one tap per k-slot, K = 196 with a K tail), a 1x1/2 conv, and a two-source Linear without
    gather; each against torch and against the general loaders (RPG_TUNE_FAST_LOADER = 0)."""
    from relpose_gnn_amd import ops
    convs = [  # n, h, w, cin, cout, k, stride, pad, residual
        (3, 13, 17, 64, 72, 3, 2, 1, True),
        (2, 37, 29, 4, 64, 7, 2, 3, False),
        (5, 9, 11, 32, 40, 1, 2, 0, False),
    ]
    try:
        ops.set_tuning(ops.TUNE_BK, bk)
        ops.set_tuning(ops.TUNE_TILE, tile)
        ops.set_tuning(ops.TUNE_STREAMK, streamk)
        for ci, (n, h, w, cin, cout, k, st, pad, res) in enumerate(convs):
            x = _rand(n, cin, h, w, seed=40 + ci)
            wt = _rand(cout, cin, k, k, seed=50 + ci, scale=(2.0 / (cin * k * k)) ** 0.5)
            scale = torch.rand(cout, generator=torch.Generator().manual_seed(60 + ci)) + 0.5
            shift = _rand(cout, seed=70 + ci, scale=0.1)
            ref = F.conv2d(x, wt, None, stride=st, padding=pad) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
            r = _rand(*ref.shape, seed=80 + ci) if res else None
            if res:
                ref = ref + r
            ref = F.relu(ref)
            args = (x.permute(0, 2, 3, 1).contiguous().to(dev), wt.permute(0, 2, 3, 1).contiguous().to(dev), scale.to(dev),
                    shift.to(dev), None if r is None else r.permute(0, 2, 3, 1).contiguous().to(dev))
            y = ops.conv2d_bn_act_nhwc(*args, stride=st, pad=pad, relu=True)
            assert rel_err(y.cpu().permute(0, 3, 1, 2), ref) < TOL, (ci, "vs torch")
            ops.set_tuning(ops.TUNE_FAST_LOADER, 0)
            y0 = ops.conv2d_bn_act_nhwc(*args, stride=st, pad=pad, relu=True)
            ops.set_tuning(ops.TUNE_FAST_LOADER, 1)
            assert rel_err(y.cpu(), y0.cpu()) < TOL, (ci, "vs general loader")
        m, widths, n_out = 333, (64, 96), 100              # K = 160: segments are multiples of 32
        srcs = [(_rand(m, wd, seed=90 + i).to(dev), None) for i, wd in enumerate(widths)]
        k = sum(widths)
        wl, bias, res = _rand(n_out, k, seed=95, scale=k ** -0.5), _rand(n_out, seed=96), _rand(m, n_out, seed=97)
        ref = F.relu(F.linear(torch.cat([a.cpu() for a, _ in srcs], 1), wl, bias) + res)
        out = ops.linear_gather(srcs, wl.to(dev), bias.to(dev), m, res.to(dev), True)
        assert rel_err(out.cpu(), ref) < TOL
    finally:
        ops.set_tuning(ops.TUNE_BK, 0)
        ops.set_tuning(ops.TUNE_TILE, -1)
        ops.set_tuning(ops.TUNE_STREAMK, 1)
        ops.set_tuning(ops.TUNE_FAST_LOADER, 1)


@pytest.mark.parametrize("sizes,k,d", [((8, 8, 8), 4, 2048), ((8, 9, 3, 1), 4, 64), ((40,), 7, 128)])
def test_knn_graph(dev, sizes, k, d):
    """rpg_knn_graph_f32 vs the oracle's restatement of torch_cluster.knn_graph (ragged graphs, graphs smaller than k+1)."""
    from relpose_gnn_amd import ops
    from oracle.posenet_ref import knn_graph
    n = sum(sizes)
    x = _rand(n, d, seed=31)
    b = torch.cat([torch.full((s,), i, dtype=torch.int64) for i, s in enumerate(sizes)])
    ref = knn_graph(x, k, b)
    got = ops.knn_graph(x.to(dev), k, b.to(dev)).cpu()
    assert torch.equal(got, ref)
    if len(sizes) == 1:
        assert torch.equal(ops.knn_graph(x.to(dev), k, None).cpu(), ref)


@pytest.mark.parametrize("n,h,w,cin,cout,res,relu", [
    (2, 9, 11, 8, 16, False, True),        # ragged width (Tw=3, last tile 3 px), Cin < K-step (several kernel rows per step)
    (1, 16, 16, 24, 72, True, True),       # Cout not a multiple of the 64-channel tile, residual
    (3, 5, 4, 16, 64, True, False),        # exactly one tile per row, no activation
    (8, 56, 56, 64, 64, True, True),       # layer1 shape
    (4, 28, 28, 128, 128, False, True),    # layer2
    (4, 14, 14, 256, 256, True, True),     # layer3: 14 -> 4 tiles per row (12.5 % padding)
    (4, 7, 7, 512, 512, True, True),       # layer4: 7 -> 2 tiles per row
    (2, 256 // 8, 341 // 8 + 1, 128, 128, True, True),   # odd width from the 256x341 evaluation shape (32x43)
    (40, 12, 9, 20, 68, True, True),       # many images per 128-tile workgroup, channel tail (Cin % 16 = 4), odd K-step count
])
@pytest.mark.parametrize("kernel", [2, 3], ids=["wave4", "wave8"])
def test_conv3x3_winograd(dev, n, h, w, cin, cout, res, relu, kernel):
    """1-D Winograd F(4,3) convolution vs F.conv2d; tolerance 2e-5 (the transforms cost ~2.5x the rounding error of
    the direct kernel per layer: measured in tools, still 5x below the per-op bar used elsewhere x 2)."""
    from relpose_gnn_amd import ops
    x = _rand(n, cin, h, w, seed=1)
    wt = _rand(cout, cin, 3, 3, seed=2, scale=(2.0 / (cin * 9)) ** 0.5)
    scale = torch.rand(cout, generator=torch.Generator().manual_seed(3)) + 0.5
    shift = _rand(cout, seed=4, scale=0.1)
    ref = F.conv2d(x, wt, None, stride=1, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    r = None
    if res:
        r = _rand(*ref.shape, seed=5)
        ref = ref + r
    if relu:
        ref = F.relu(ref)
    w_ohwi = wt.permute(0, 2, 3, 1).contiguous().to(dev)
    u = ops.wino43_transform_weights(w_ohwi)
    # weight transform against a float64 statement of U = G g
    G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                      [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)
    u_ref = torch.einsum("xj,ohjc->xohc", G, wt.permute(0, 2, 3, 1).double()).float()
    assert torch.allclose(u.cpu(), u_ref, rtol=2e-7, atol=1e-9)
    ops.set_tuning(ops.TUNE_WINOGRAD, kernel)          # 2: the 4-wave / 64-tile kernel, 3: the 8-wave / 128-tile one
    try:
        y = ops.conv3x3_wino43_bn_act_nhwc(x.permute(0, 2, 3, 1).contiguous().to(dev), u, scale.to(dev), shift.to(dev),
                                           None if r is None else r.permute(0, 2, 3, 1).contiguous().to(dev), relu=relu)
    finally:
        ops.set_tuning(ops.TUNE_WINOGRAD, 1)
    err = rel_err(y.cpu().permute(0, 3, 1, 2), ref)
    assert err < 2e-5, err


@pytest.mark.parametrize("n,h,w,cin,cout,res", [
    (100, 24, 30, 32, 96, True),           # 300 tiles: no split possible (one channel-block pair), ragged Cout, Tw = 8
    (37, 33, 41, 64, 132, True),           # ragged everything: Tw = 11 (3-px last tile), Cout tail of 4 channels, M % 128 != 0
    (60, 28, 28, 128, 128, False),         # 184 m-tiles x 2: tail of 112 tiles -> 2 parts each
    (33, 56, 56, 64, 64, True),            # 809 tiles: 3 whole rounds + 41 tail tiles x 2 parts
    (9, 50, 47, 96, 64, True),             # Cin = 96: six channel blocks, odd width
    (140, 32, 32, 48, 64, True),           # Cin = 48: THREE channel blocks = 9 K steps per tile: the LDS image roles swap after
                                           # every tile; 280 tiles, tail of 24 -> 3 parts of one channel block each
    (70, 32, 32, 16, 128, False),          # Cin = 16: one channel block (3 K steps) per tile, no split possible
])
def test_conv3x3_winograd_persistent(dev, n, h, w, cin, cout, res):
    """The persistent 8-wave kernel (more tiles than CUs, Cin % 16 == 0: loads pipelined across the tiles of a workgroup)
    against F.conv2d, with and without the split-K tail, and against the one-workgroup-per-tile kernel to 1e-6 (whole
    tiles run the same K walk in both; only the output transform is associated differently)."""
    from relpose_gnn_amd import ops
    x = _rand(n, cin, h, w, seed=11)
    wt = _rand(cout, cin, 3, 3, seed=12, scale=(2.0 / (cin * 9)) ** 0.5)
    scale = torch.rand(cout, generator=torch.Generator().manual_seed(13)) + 0.5
    shift = _rand(cout, seed=14, scale=0.1)
    ref = F.conv2d(x, wt, None, stride=1, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    r = _rand(*ref.shape, seed=15) if res else None
    ref = F.relu(ref + r) if res else F.relu(ref)
    u = ops.wino43_transform_weights(wt.permute(0, 2, 3, 1).contiguous().to(dev))
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    rd = None if r is None else r.permute(0, 2, 3, 1).contiguous().to(dev)
    out = {}
    for persist in (1, 0):
        for split in (1, 0):
            ops.set_tuning(ops.TUNE_WINOGRAD, 3)
            ops.set_tuning(ops.TUNE_WINO_PERSIST, persist)
            ops.set_tuning(ops.TUNE_WINO_SPLIT, split)
            try:
                out[(persist, split)] = ops.conv3x3_wino43_bn_act_nhwc(xd, u, scale.to(dev), shift.to(dev), rd, relu=True).cpu()
            finally:
                ops.set_tuning(ops.TUNE_WINOGRAD, 1)
                ops.set_tuning(ops.TUNE_WINO_PERSIST, 1)
                ops.set_tuning(ops.TUNE_WINO_SPLIT, 1)
    for k, y in out.items():
        e = rel_err(y.permute(0, 3, 1, 2), ref)
        assert e < 2e-5, (k, e)
    assert rel_err(out[(1, 0)], out[(0, 0)]) < 1e-6


def test_composite_abi_error_codes(dev):
    """rpg_resnet_forward_f32 / rpg_gnn_forward_f32 reject a wrong tensor table or a short workspace with status codes
    (no launch, no exception across the ABI)."""
    from relpose_gnn_amd import _lib
    lib = _lib.lib()
    planes, blocks = _lib.int_array([8, 16, 32, 64]), _lib.int_array([1, 1, 1, 1])
    x = torch.zeros(2, 3, 32, 32, device=dev)
    feat = torch.zeros(2, 64, device=dev)
    ws = torch.zeros(1 << 20, dtype=torch.uint8, device=dev)
    junk = torch.zeros(16, device=dev)
    ptrs = _lib.ptr_array([junk.data_ptr()] * 5)
    rc = lib.rpg_resnet_forward_f32(ptrs, 5, blocks, planes, 64, x.data_ptr(), 2, 32, 32, feat.data_ptr(), ws.data_ptr(),
                                    ws.numel(), None)
    assert rc == _lib.RPG_ERR_BAD_ARG                                   # wrong number of tensors
    n_t = 4 + 8 * 4 + 4 * 3 + 2
    ptrs = _lib.ptr_array([junk.data_ptr()] * n_t)
    rc = lib.rpg_resnet_forward_f32(ptrs, n_t, blocks, planes, 64, x.data_ptr(), 2, 32, 32, feat.data_ptr(), ws.data_ptr(), 16, None)
    assert rc == _lib.RPG_ERR_WORKSPACE
    ei = torch.zeros(2, 4, dtype=torch.int64, device=dev)
    out = torch.zeros(64, device=dev)
    st = torch.zeros(1, dtype=torch.int32, device=dev)
    g = _lib.ptr_array([junk.data_ptr()] * 22)
    rc = lib.rpg_gnn_forward_f32(g, 22, feat.data_ptr(), ei.data_ptr(), ei.data_ptr() + 32, 0, 2, 4, 64, 2, out.data_ptr(),
                                 out.data_ptr(), None, None, st.data_ptr(), ws.data_ptr(), 16, None)
    assert rc == _lib.RPG_ERR_WORKSPACE
    rc = lib.rpg_gnn_forward_f32(g, 21, feat.data_ptr(), ei.data_ptr(), ei.data_ptr() + 32, 0, 2, 4, 64, 2, out.data_ptr(),
                                 out.data_ptr(), None, None, st.data_ptr(), ws.data_ptr(), ws.numel(), None)
    assert rc == _lib.RPG_ERR_BAD_ARG
    with pytest.raises(_lib.RpgError):
        _lib.check(_lib.RPG_ERR_WORKSPACE, "x")


# ---- in-kernel combine of split-K / stream-K partial tiles (round 4) ------------------------------------------------------
def _with_fixup_mode(mode, fn):
    from relpose_gnn_amd import ops
    ops.set_tuning(ops.TUNE_INKERNEL_FIXUP, mode)
    try:
        return fn()
    finally:
        ops.set_tuning(ops.TUNE_INKERNEL_FIXUP, 0)


@pytest.mark.parametrize("n,h,c,res", [
    (8, 56, 64, True),        # one 8-node graph, layer 1: 49 tiles, all split 4 ways (the reference's batch_size=1 loop, test.py:192)
    (8, 28, 128, True),       # layer 2 of one graph: 26 tiles x 8 parts
    (8, 14, 256, False),      # layer 3: 16 tiles x 16 parts -- above the in-kernel limit: separate fix-up launch unless mode = 32
    (128, 7, 512, True),      # layer 4 of a 16-graph half batch: 112 tiles x 2 parts (the configs[1] two-stream geometry)
    (5, 37, 68, True),        # ragged: 37 -> 40 wide rows, Cout % 64 != 0, partial last tile
])
def test_winograd_split_parts_combined_by_last_arriver(dev, n, h, c, res):
    """The tail tiles of wino43_conv8_kernel are cut along K; the workgroup that stores a tile's LAST part adds the parts in
    k order and applies BatchNorm / residual / ReLU itself (per-tile arrival counter, RPG_TUNE_INKERNEL_FIXUP bit 1) instead of
    a wino43_fixup_kernel launch (0, the default: the in-kernel form measured slower).  Asserted: same result as the fix-up launch (same sums in the same order; 1e-6 allows
    a different FMA contraction), both <= 2e-5 of F.conv2d, and BITWISE run-to-run determinism over 6 runs -- whichever
    workgroup arrives last, the bits are the same.  Mode 6 (= 2 + 4) pushes the 16-part shape through the in-kernel path too."""
    from relpose_gnn_amd import ops
    g = torch.Generator().manual_seed(500 + h + c)
    x = torch.randn(n, c, h, h + (3 if c == 68 else 0), generator=g)
    wt = torch.randn(c, c, 3, 3, generator=g) * (2.0 / (c * 9)) ** 0.5
    scale, shift = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1
    r = torch.randn(x.shape, generator=g) if res else None
    ref = F.conv2d(x, wt, None, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    ref = F.relu(ref + r) if res else F.relu(ref)
    nh = lambda t: None if t is None else t.permute(0, 2, 3, 1).contiguous().to(dev)
    u = ops.wino43_transform_weights(nh(wt))
    xd, rd, sc, sh = nh(x), nh(r), scale.to(dev), shift.to(dev)
    ops.set_tuning(ops.TUNE_WINOGRAD, 3)
    try:
        run = lambda: ops.conv3x3_wino43_bn_act_nhwc(xd, u, sc, sh, rd, relu=True)
        y_launch = _with_fixup_mode(0, run)
        outs = {mode: [_with_fixup_mode(mode, run) for _ in range(6 if mode == 2 else 3)] for mode in (2, 6)}
    finally:
        ops.set_tuning(ops.TUNE_WINOGRAD, 1)
    assert rel_err(y_launch.cpu().permute(0, 3, 1, 2), ref) < 2e-5
    for mode, ys in outs.items():
        assert all(torch.equal(y, ys[0]) for y in ys[1:]), f"mode {mode}: not deterministic"
        assert rel_err(ys[0], y_launch) < 1e-6, mode
        assert rel_err(ys[0].cpu().permute(0, 3, 1, 2), ref) < 2e-5, mode


@pytest.mark.parametrize("m,k,n_out,gather,res", [
    (1792, 2048, 2048, False, True),      # a GNN edge Linear at configs[1]: 224 tiles of 128 x 128 on 512 slots, every tile stream-K split
    (256, 4096, 2048, False, False),      # mlp_updating.0 on the node rows: 32 tiles, K cut 8+ ways
    (448, 6144, 2048, True, True),        # edge_mlp.0 in the reference formulation (3 gathered sources) for one 8-graph group
    (56, 2048, 768, False, False),        # one graph's g|theta|phi projection
    (1000, 2304, 260, False, True),       # ragged M / N
])
def test_streamk_partial_tiles_combined_by_last_arriver(dev, m, k, n_out, gather, res):
    """Same for the stream-K remainder of the fp32 GEMM engine (gemm_streamk_kernel): in-kernel combine (bit 0) vs the
    streamk_fixup_kernel launch (default), vs torch on the CPU, and bitwise determinism over 6 runs."""
    from relpose_gnn_amd import ops
    g = torch.Generator().manual_seed(900 + m)
    w = torch.randn(n_out, k, generator=g) * (1.0 / k) ** 0.5
    b = torch.randn(n_out, generator=g) * 0.1
    r = torch.randn(m, n_out, generator=g) if res else None
    if gather:
        rows = 64
        a = torch.randn(rows, k // 3, generator=g)
        idx = [torch.randint(0, rows, (m,), generator=g) for _ in range(3)]
        ref = torch.cat([a[i] for i in idx], 1) @ w.t() + b
        src = [(a.to(dev), i.to(dev)) for i in idx]
    else:
        a = torch.randn(m, k, generator=g)
        ref = a @ w.t() + b
        src = [(a.to(dev), None)]
    ref = F.relu(ref + r) if res else ref
    wd, bd, rd = w.to(dev), b.to(dev), None if r is None else r.to(dev)
    run = lambda: ops.linear_gather(src, wd, bd, m, residual=rd, relu=res)
    y_launch = _with_fixup_mode(0, run)
    ys = [_with_fixup_mode(1, run) for _ in range(6)]
    assert rel_err(y_launch.cpu(), ref) < 1e-5
    assert all(torch.equal(y, ys[0]) for y in ys[1:]), "not deterministic"
    assert rel_err(ys[0], y_launch) < 1e-6 and rel_err(ys[0].cpu(), ref) < 1e-5


@pytest.mark.parametrize("m,k,n_out,res,relu", [(1792, 2048, 2048, False, True), (896, 2048, 2048, True, False), (3584, 512, 1024, True, True),
                                                 (1792, 96, 2048, False, False), (896, 2080, 2048, False, True)])
def test_linear_exact_fit_112_row_tiles(dev, m, k, n_out, res, relu):
    """Round 5: Linears with M = 7 * 2^k rows (the GNN's edge GEMMs: 56 edges x graphs) on 112 x 64 tiles of
    v_mfma_f32_16x16x4_f32 -- no stream-K split, no fix-up launch (RPG_TUNE_LIN112 = 1, csrc/gemm_f32.hip linear112_kernel;
    reference op: nn.Linear of my_gnn_layer.py:236-239,304-311).  Against F.linear on the CPU (1e-5 like every fp32 op test)
    and against the 32x32x2 tile engine on the same operands (same bar: the two kernels sum in different orders, so they agree
    to rounding, not bit for bit): K % 32 tails that are not a multiple of the fold length,
    K = 96 (three steps: odd step count), residual rows, ReLU, 1 / 2 / 4 tiles per CU."""
    from relpose_gnn_amd import ops
    a = _rand(m, k, seed=31)
    wt = _rand(n_out, k, seed=32, scale=k ** -0.5)
    bias = _rand(n_out, seed=33, scale=0.1)
    ref = F.linear(a, wt, bias)
    r = None
    if res:
        r = _rand(m, n_out, seed=34)
        ref = ref + r
    if relu:
        ref = F.relu(ref)
    outs = {}
    try:
        for mode in (3, 1, 0):                               # eight-wave form where about one tile per CU / four-wave form only / engine
            ops.set_tuning(ops.TUNE_LIN112, mode)
            outs[mode] = ops.linear_gather([(a.to(dev), None)], wt.to(dev), bias.to(dev), m, None if r is None else r.to(dev), relu).cpu()
    finally:
        ops.set_tuning(ops.TUNE_LIN112, 3)
    for mode in (3, 1, 0):
        assert rel_err(outs[mode], ref) < TOL, (mode, rel_err(outs[mode], ref))
    assert rel_err(outs[1], outs[0]) < TOL and rel_err(outs[3], outs[0]) < TOL


@pytest.mark.parametrize("m,nodes,k,pad_a,pad_r", [(1792, 256, 2048, 64, 0), (896, 128, 2048, 0, 64), (1792, 256, 1024, 32, 32)])
def test_linear_exact_fit_112_epilogue_and_strided_a(dev, m, nodes, k, pad_a, pad_r):
    """The exact-fit 112-row kernels share the tile engine's epilogue code (streamk_finish_quad) and its buffer-offset
    arithmetic; until round 6 only the whole GNN forward exercised them there.  Through rpg_linear_gather_ex_f32:
    a plain A operand that is a column block of a wider tensor (row pitch > K), TWO gathered residual rows per output row (the
    node terms of the split edge Linears, my_gnn_layer.py:236-239: W_src x[src] + W_dst x[dst]) out of tensors whose pitch may exceed
    n_out, bias, ReLU, and the max(out, 0) second output -- on the four-wave kernel, its eight-wave form and the engine."""
    from relpose_gnn_amd import ops
    n_out = 2048
    a_wide = _rand(m, k + pad_a, seed=41)
    wt = _rand(n_out, k, seed=42, scale=k ** -0.5)
    bias = _rand(n_out, seed=43, scale=0.1)
    r1 = _rand(nodes, n_out + pad_r, seed=44)
    r2 = _rand(nodes, n_out + pad_r, seed=45)
    g = torch.Generator().manual_seed(46)
    i1 = torch.randint(0, nodes, (m,), generator=g)
    i2 = torch.randint(0, nodes, (m,), generator=g)
    pre = F.linear(a_wide[:, :k], wt, bias) + r1[i1, :n_out] + r2[i2, :n_out]
    outs = {}
    try:
        for mode in (3, 1, 0):
            ops.set_tuning(ops.TUNE_LIN112, mode)
            for relu in (False, True):
                o, o_relu = ops.linear_gather_ex([(a_wide.to(dev), None)], wt.to(dev), bias.to(dev), m, r1.to(dev), i1.to(dev), r2.to(dev),
                                                 i2.to(dev), relu=relu, want_relu_copy=True, widths=[k])
                outs[(mode, relu)] = (o.cpu(), o_relu.cpu())
    finally:
        ops.set_tuning(ops.TUNE_LIN112, 3)
    for (mode, relu), (o, o_relu) in outs.items():
        want = F.relu(pre) if relu else pre
        assert rel_err(o, want) < TOL, (mode, relu, rel_err(o, want))
        assert torch.equal(o_relu, F.relu(o)) or rel_err(o_relu, F.relu(pre)) < TOL, (mode, relu)
    # plain (un-gathered) residual through the same entry point == rpg_linear_gather_f32
    ops.set_tuning(ops.TUNE_LIN112, 3)
    res = _rand(m, n_out, seed=47)
    o = ops.linear_gather_ex([(a_wide.to(dev), None)], wt.to(dev), bias.to(dev), m, res.to(dev), relu=True, widths=[k]).cpu()
    assert rel_err(o, F.relu(F.linear(a_wide[:, :k], wt, bias) + res)) < TOL
