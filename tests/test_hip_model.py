"""GPU parity of the whole hot path through the PoseNetX_R2 mirror (which calls the C ABI): HIP forward vs
(a) the golden vectors written by the reference itself (tests/golden/make_golden.py) and (b) the CPU oracle run live on
the same deterministic weights/inputs.  Tolerance: 1e-4 relative (max-norm), the north-star's fp32 bar."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _build(D, img_h, planes, blocks, dev, seed=1, droprate=0.0):
    import relpose_gnn_amd.synth as S
    from relpose_gnn_amd.posenet import PoseNetX_R2
    from relpose_gnn_amd.resnet import ResNet
    m = PoseNetX_R2(ResNet(blocks, planes), droprate=droprate, pretrained=False, feat_dim=D, edge_feat_dim=D, node_dim=D,
                    input_img_height=img_h, use_gnn=True, knn=-1, use_AP=True, gnn_recursion=2)
    sd = S.synth_state_dict(S.posenet_r2_param_shapes(D, D, D, planes, blocks), seed=seed)
    m.load_state_dict(sd)
    return m.to(dev).eval(), sd


def _report(name, ea, er):
    """Append the measured parity numbers to gpurun_out/parity_report.jsonl (scratch, copied to profiles/ by hand)."""
    import json
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "parity_report.jsonl"), "a") as f:
            f.write(json.dumps({"case": name, "abs_pose_rel_err": ea, "rel_pose_rel_err": er}) + "\n")


def _data(x, n_nodes, dev):
    from relpose_gnn_amd.graph import fc_batch
    return fc_batch(x, n_nodes).to(dev)


def test_small_model_vs_golden_g4(dev, golden_dir):
    import relpose_gnn_amd.synth as S
    m, _ = _build(64, 32, (8, 16, 32, 64), (1, 1, 1, 1), dev)
    x = S.synth_images(16, 32, 40, seed=3)
    a, r, ei = m(_data(x, 8, dev))
    g = np.load(os.path.join(golden_dir, "g4_full_small.npz"))
    _report("g4_small_encoder_D64_vs_reference_golden", rel_err(a.cpu(), g["abs"]), rel_err(r.cpu(), g["rel"]))
    assert rel_err(a.cpu(), g["abs"]) < TOL and rel_err(r.cpu(), g["rel"]) < TOL
    assert ei.shape == (2, 112)


def test_reference_written_files_to_forward_g9(dev, golden_dir):
    """file -> ``io.load_checkpoint_state_dict`` -> ``load_state_dict`` -> forward == G4, the path of
    testing/test.py:347-348.  The checkpoint was written by the reference's own ``save_checkpoint`` (utils/utils.py:22-31,
    make_golden.py G9) and the graph sample has the ``Data`` pickle layout of dataset_7Scenes_multi.py:437-446."""
    from relpose_gnn_amd import io as rio
    from relpose_gnn_amd.graph import Batch
    from relpose_gnn_amd.posenet import PoseNetX_R2
    from relpose_gnn_amd.resnet import ResNet
    m = PoseNetX_R2(ResNet((1, 1, 1, 1), (8, 16, 32, 64)), droprate=0.0, pretrained=False, feat_dim=64, edge_feat_dim=64,
                    node_dim=64, input_img_height=32, use_gnn=True, knn=-1, use_AP=True, gnn_recursion=2)
    checkpoint_sd = rio.load_checkpoint_state_dict(os.path.join(golden_dir, "epoch_199.pth.tar"))
    missing = m.load_state_dict(checkpoint_sd)                   # strict: every key of the reference's file is consumed
    assert not missing.missing_keys and not missing.unexpected_keys
    m = m.to(dev).eval()
    sample = rio.load_graph(rio.processed_files(golden_dir)[0])
    a, r, ei = m(Batch.from_data_list([sample]).to(dev))
    g = np.load(os.path.join(golden_dir, "g4_full_small.npz"))   # G4 = 2 graphs; the sample is graph 0 (graphs are independent)
    ea = float((a.cpu() - torch.from_numpy(g["abs"][:8])).abs().max() / np.abs(g["abs"]).max())
    er = float((r.cpu() - torch.from_numpy(g["rel"][:56])).abs().max() / np.abs(g["rel"]).max())
    _report("g9_reference_written_checkpoint_and_sample_to_forward_vs_g4", ea, er)
    assert ea < TOL and er < TOL and ei.shape == (2, 56), (ea, er)


def test_resnet34_64px_vs_golden_g4b(dev, golden_dir):
    import relpose_gnn_amd.synth as S
    m, _ = _build(64, 64, (64, 128, 256, 512), (3, 4, 6, 3), dev)
    x = S.synth_images(4, 64, 64, seed=4)
    a, r, _ = m(_data(x, 4, dev))
    g = np.load(os.path.join(golden_dir, "g4b_resnet34_64px.npz"))
    feat = m.feature_extractor.__class__.forward(m.feature_extractor, x.view(4, 3, 64, 64).to(dev))
    assert rel_err(feat.cpu(), g["feat"]) < TOL
    _report("g4b_resnet34_64px_D64_vs_reference_golden", rel_err(a.cpu(), g["abs"]), rel_err(r.cpu(), g["rel"]))
    assert rel_err(a.cpu(), g["abs"]) < TOL and rel_err(r.cpu(), g["rel"]) < TOL


def test_full_r3_224_vs_golden_g5_and_oracle(dev, golden_dir):
    """BASELINE.json config 0 shape: one 4-node FC graph, 224x224, D=2048 (the reference's CPU-runnable case)."""
    import relpose_gnn_amd.synth as S
    from oracle import posenet_ref as O
    m, sd = _build(2048, 224, (64, 128, 256, 512), (3, 4, 6, 3), dev)
    x = S.synth_images(4, 224, 224, seed=5)
    a, r, _ = m(_data(x, 4, dev))
    g = np.load(os.path.join(golden_dir, "g5_full_r3_224.npz"))
    ea, er = rel_err(a.cpu(), g["abs"]), rel_err(r.cpu(), g["rel"])
    _report("g5_R3_D2048_224px_4node_vs_reference_golden", ea, er)
    assert ea < TOL and er < TOL, (ea, er)
    # second input through the live oracle: 2 graphs x 8 nodes (batched index offsets at full width)
    x2 = S.synth_images(16, 224, 224, seed=6)
    d2 = _data(x2, 8, dev)
    a2, r2, _ = m(d2)
    oa, orr, _ = O.posenet_forward(sd, x2, d2.edge_index.cpu(), 224, 2)
    ea, er = rel_err(a2.cpu(), oa), rel_err(r2.cpu(), orr)
    _report("R3_D2048_224px_2x8node_vs_live_oracle", ea, er)
    assert ea < TOL and er < TOL, (ea, er)


def test_gnn_stages_vs_golden_g2_g3(dev, golden_dir):
    """GNN-only at D=64 through the fine-grained C-ABI ops, stage by stage (B=1 and B=3 graphs)."""
    import relpose_gnn_amd.synth as S
    from relpose_gnn_amd import ops
    from relpose_gnn_amd.params import pack_gnn
    from oracle.posenet_ref import batch_edge_index
    sd = S.synth_state_dict(S.posenet_r2_param_shapes(64, 64, 64, (8, 16, 32, 64), (1, 1, 1, 1)), seed=1)
    t = [v.to(dev) for v in pack_gnn(sd)]
    for tag, B in (("g2", 1), ("g3", 3)):
        g = np.load(os.path.join(golden_dir, f"{tag}_gnn_d64_b{B}.npz"))
        x = S.hash_normal(f"{tag}.feat", (8 * B, 64), 1.0, 0.0, seed=2).to(dev)
        ei = batch_edge_index(8, B).to(dev)
        n, e = 8 * B, 56 * B
        gp = ops.graph_prepare(ei, n)
        src, dst, lo, hi = gp["ends"][0], gp["ends"][1], gp["ends"][2], gp["ends"][3]
        ecur = ops.linear_gather([(x, lo), (x, hi)], t[0], t[1], e, relu=True)
        assert rel_err(ecur.cpu(), g["stage_proj_edge"]) < TOL
        for r in range(2):
            hid = ops.linear_gather([(x, src), (x, dst), (ecur, None)], t[2], t[3], e, relu=True)
            enew = ops.linear_gather([(hid, None)], t[4], t[5], e)
            assert rel_err(enew.cpu(), g[f"stage_r{r}.edge_update"]) < TOL
            hid = ops.linear_gather([(x, src), (enew, None)], t[6], t[7], e, relu=True)
            msg = ops.linear_gather([(hid, None)], t[8], t[9], e)
            assert rel_err(msg.cpu(), g[f"stage_r{r}.msg_mlp"]) < TOL
            y = ops.attention_rows(ops.linear_gather([(msg, None)], t[10], t[11], e))
            att = ops.linear_gather([(y, None)], t[12], t[13], e, residual=msg)
            assert rel_err(att.cpu(), g[f"stage_r{r}.att"]) < TOL
            agg = ops.scatter_mean(att, gp["rowptr"], gp["perm"], n)
            assert rel_err(agg.cpu(), g[f"stage_r{r}.aggregate"]) < TOL
            nh = ops.linear_gather([(x, None), (agg, None)], t[14], t[15], n, relu=True)
            xn = ops.linear_gather([(nh, None)], t[16], t[17], n)
            assert rel_err(xn.cpu(), g[f"stage_r{r}.node_update"]) < TOL
            x, ecur = torch.relu(xn), torch.relu(enew)
        assert rel_err(ops.pose_heads(x, t[18], t[19]).cpu(), g["abs"]) < TOL
        assert rel_err(ops.pose_heads(ecur, t[20], t[21]).cpu(), g["rel"]) < TOL


def test_module_contract(dev):
    """Return types / errors the evaluation script relies on (testing/test.py:211-229)."""
    import relpose_gnn_amd.synth as S
    m, _ = _build(64, 32, (8, 16, 32, 64), (1, 1, 1, 1), dev)
    x = S.synth_images(8, 32, 40, seed=3)
    d = _data(x, 8, dev)
    a, r, ei = m(d)
    assert a.shape == (8, 6) and r.shape == (56, 6) and ei is d.edge_index and not a.requires_grad
    assert a.device.type == "cuda" and int((ei.cpu()[1] == 0).nonzero()[0]) == 28      # reference edge (1 -> 0)
    with pytest.raises(RuntimeError):
        m(_data(x, 8, torch.device("cpu")))                 # no CPU fallback
    mixed = _data(x, 8, dev)
    mixed.edge_index = mixed.edge_index.cpu()
    with pytest.raises(RuntimeError):
        m(mixed)                                            # edge_index left on the host
    half = _data(x, 8, dev)
    half.x = half.x.half()
    with pytest.raises(TypeError):
        m(half)
    # a node id out of range: the reference's indexing raises IndexError.  Every call is validated on the device; the
    # default mode reports without blocking inside forward (at check_edge_index() / the next call), "sync" at the call
    bad = _data(x, 8, dev)
    bad.edge_index = bad.edge_index.clone()
    bad.edge_index[0, 5] = 99
    assert m.index_check == "deferred"
    m(bad)                                                  # returns (the bad edge is clamped and left out on the device)
    with pytest.raises(IndexError):
        m.check_edge_index()
    m.check_edge_index()                                    # reported once; the counter is cleared
    m(bad)
    torch.cuda.synchronize()
    with pytest.raises(IndexError):
        m(d)                                                # the NEXT call reports it too
    a_ok, _, _ = m(d)
    m.check_edge_index()
    assert torch.equal(a_ok, a)
    # the same storage refilled with different (bad) indices is validated again: nothing is cached on data_ptr
    good = _data(x, 8, dev)
    m(good)
    m.check_edge_index()
    good.edge_index[1, 7] = -3
    m(good)
    with pytest.raises(IndexError):
        m.check_edge_index()
    m.index_check = "sync"
    with pytest.raises(IndexError):
        m(bad)
    a_ok, _, _ = m(d)
    assert torch.equal(a_ok, a)
    m.index_check = "deferred"
    # dropout-faithful path runs and changes the output (always-on dropout of the reference)
    m.droprate = 0.5
    a2, r2, _ = m(d)
    m.droprate = 0.0
    assert a2.shape == (8, 6) and not torch.allclose(a2, a)
    # determinism of the droprate=0 path
    a3, r3, _ = m(d)
    assert torch.equal(a3, a) and torch.equal(r3, r)


@pytest.mark.parametrize("use_AP", [True, False])
def test_dropout_always_on_with_seeded_mask(dev, use_AP):
    """A11 (posenet.py:1073-1075): F.dropout(x, p) / F.dropout(edge_feat, p) with training=True regardless of eval().
    The random draw cannot match across devices, but its arithmetic can be pinned: with the GPU generator re-seeded, the
    masks the module drew are reproduced by F.dropout on tensors of the same shapes in the same order, and the oracle's
    heads on the identically masked features must equal the module's outputs.  Also: keep probability and 1/(1-p) scale."""
    import relpose_gnn_amd.synth as S
    from oracle import posenet_ref as O
    from relpose_gnn_amd.posenet import PoseNetX_R2
    from relpose_gnn_amd.resnet import ResNet
    D, p = 64, 0.5
    planes, blocks = (8, 16, 32, 64), (1, 1, 1, 1)
    m = PoseNetX_R2(ResNet(blocks, planes), droprate=p, pretrained=False, feat_dim=D, edge_feat_dim=D, node_dim=D,
                    input_img_height=32, use_gnn=True, knn=-1, use_AP=use_AP, gnn_recursion=2)
    sd = S.synth_state_dict(S.posenet_r2_param_shapes(D, D, D, planes, blocks, use_AP=use_AP), seed=1)
    m.load_state_dict(sd)
    m = m.to(dev).eval()                                    # eval() does NOT switch the reference's dropout off
    x = S.synth_images(24, 32, 40, seed=3)
    d = _data(x, 8, dev)
    n, e = 24, 3 * 56
    torch.cuda.manual_seed(1234)
    a, r, _ = m(d)
    torch.cuda.manual_seed(1234)
    ones_n, ones_e = torch.ones(n, D, device=dev), torch.ones(e, D, device=dev)
    mask_n = torch.nn.functional.dropout(ones_n, p=p)       # same generator state, same shapes, same order as forward
    mask_e = torch.nn.functional.dropout(ones_e, p=p)
    vals = torch.unique(torch.cat([mask_n.flatten(), mask_e.flatten()])).tolist()
    assert vals == [0.0, 1.0 / (1.0 - p)]                   # dropped -> 0, kept -> scaled by 1/(1-p)
    keep = float((mask_e > 0).float().mean())
    assert abs(keep - (1.0 - p)) < 0.02, keep               # 10752 draws: 4 sigma = 0.02
    oa, orr, _ = O.posenet_forward(sd, x, d.edge_index.cpu(), 32, 2, use_AP=use_AP, node_mask=mask_n.cpu(),
                                   edge_mask=mask_e.cpu())
    ea, er = rel_err(a.cpu(), oa), rel_err(r.cpu(), orr)
    _report(f"dropout_seeded_mask_use_AP_{use_AP}", ea, er)
    assert ea < TOL and er < TOL, (ea, er)
    a2, _, _ = m(d)                                         # a fresh draw differs (always-on)
    assert not torch.allclose(a2, a)


def test_batch_independence_full_width(dev):
    """Size-independent property at a BASELINE-sized batch (32 graphs x 8 nodes, D=2048, small images to bound the
    oracle-free check): a graph's poses do not depend on which other graphs share the batch."""
    import relpose_gnn_amd.synth as S
    from oracle import posenet_ref as O
    m, sd = _build(2048, 64, (64, 128, 256, 512), (3, 4, 6, 3), dev)
    x = S.synth_images(8 * 32, 64, 64, seed=8)
    a, r, _ = m(_data(x, 8, dev))
    a1, r1, _ = m(_data(x[8 * 5: 8 * 6], 8, dev))
    oa, orr, _ = O.posenet_forward(sd, x[8 * 5: 8 * 6], O.fc_edge_index(8), 64, 2)
    _report("R3_D2048_64px_graph5_in_batch32_vs_live_oracle", rel_err(a[40:48].cpu(), oa), rel_err(r[56 * 5: 56 * 6].cpu(), orr))
    _report("R3_D2048_64px_graph5_alone_vs_live_oracle", rel_err(a1.cpu(), oa), rel_err(r1.cpu(), orr))
    assert rel_err(a[40:48].cpu(), oa) < TOL and rel_err(a1.cpu(), oa) < TOL
    # not bit-equal: the stream-K split of the K range (hence the fp32 summation order) depends on the batch size
    ea, er = rel_err(a[40:48].cpu(), a1.cpu()), rel_err(r[56 * 5: 56 * 6].cpu(), r1.cpu())
    _report("batch_independence_32_vs_1_graphs", ea, er)
    assert ea < TOL and er < TOL, (ea, er)


def test_constructor_flags_vs_golden_g7(dev, golden_dir):
    """use_attention=True, use_AP=False, L=2 (posenet.py:961-972,1040-1041,1080-1083)."""
    import relpose_gnn_amd.synth as S
    from relpose_gnn_amd.posenet import PoseNetX_R2
    from relpose_gnn_amd.resnet import ResNet
    planes, blocks = (8, 16, 32, 64), (1, 1, 1, 1)
    m = PoseNetX_R2(ResNet(blocks, planes), droprate=0.0, pretrained=False, feat_dim=64, edge_feat_dim=64, node_dim=64,
                    input_img_height=32, use_gnn=True, use_attention=True, knn=-1, use_AP=False, gnn_recursion=2, L=2)
    m.load_state_dict(S.synth_state_dict(S.posenet_r2_param_shapes(64, 64, 64, planes, blocks, use_attention=True,
                                                                    use_AP=False, L=2), seed=7))
    m = m.to(dev).eval()
    a, r, _ = m(_data(S.synth_images(16, 32, 40, seed=3), 8, dev))
    g = np.load(os.path.join(golden_dir, "g7_flags_att_noAP_L2.npz"))
    assert a.shape == (112, 6)
    ea, er = rel_err(a.cpu(), g["abs"]), rel_err(r.cpu(), g["rel"])
    _report("g7_flags_attention_noAP_L2_vs_reference_golden", ea, er)
    assert ea < TOL and er < TOL


def test_knn_paths_vs_golden_g8(dev, golden_dir):
    """knn>0 in the constructor, forward(data, k), and both (posenet.py:1043-1050, 1088-1091)."""
    import relpose_gnn_amd.synth as S
    m, _ = _build(64, 32, (8, 16, 32, 64), (1, 1, 1, 1), dev)
    g = np.load(os.path.join(golden_dir, "g8_knn_flow.npz"))
    d = _data(S.synth_images(16, 32, 40, seed=9), 8, dev)
    for tag, knn, k in (("ctor", 3, None), ("both", 3, 2), ("fwd", -1, 2)):
        m.knn = knn
        a, r, e = m(d, k=k)
        assert np.array_equal(e.cpu().numpy(), g["ei_" + tag]), tag
        ea, er = rel_err(a.cpu(), g["abs_" + tag]), rel_err(r.cpu(), g["rel_" + tag])
        _report(f"g8_knn_{tag}_vs_reference_golden", ea, er)
        assert ea < TOL and er < TOL
    m.knn = -1


def test_eval_shape_256x341_vs_oracle(dev):
    """BASELINE.json configs[3] shape: the 7-Scenes evaluation images are 256x341 (odd width, ragged tiles everywhere),
    8-node FC graph, R3 dims; one graph through the live oracle."""
    import relpose_gnn_amd.synth as S
    from oracle import posenet_ref as O
    m, sd = _build(2048, 256, (64, 128, 256, 512), (3, 4, 6, 3), dev)
    x = S.synth_images(8, 256, 341, seed=12)
    d = _data(x, 8, dev)
    a, r, _ = m(d)
    oa, orr, _ = O.posenet_forward(sd, x, d.edge_index.cpu(), 256, 2)
    ea, er = rel_err(a.cpu(), oa), rel_err(r.cpu(), orr)
    _report("R3_D2048_256x341_8node_vs_live_oracle", ea, er)
    assert ea < TOL and er < TOL, (ea, er)


def test_eval_harness_end_to_end(dev):
    """evaluate_stream on the HIP module == oracle forward + oracle post-processing (test.py:213-276 semantics)."""
    import relpose_gnn_amd.synth as S
    from oracle import posenet_ref as O
    from relpose_gnn_amd import evaluate as E
    from relpose_gnn_amd.graph import Data, fc_edge_index
    m, sd = _build(64, 32, (8, 16, 32, 64), (1, 1, 1, 1), dev)
    graphs, ref_pred = [], []
    pm, ps = np.array([0.5, -1.0, 2.0]), np.array([2.0, 3.0, 0.5])
    for i in range(7):
        x = S.synth_images(8, 32, 40, seed=100 + i)
        y = S.hash_normal(f"eval.y{i}", (8, 6), 0.3)
        graphs.append(Data(x=x, edge_index=fc_edge_index(8), y=y))
        _, rel, _ = O.posenet_forward(sd, x, fc_edge_index(8), 32, 2)
        raw = O.query_pose_from_relative(rel.numpy().astype(np.float64), y.numpy().astype(np.float64), fc_edge_index(8).numpy())
        ref_pred.append(np.hstack((raw[:3] * ps + pm, O.qexp(raw[3:]))))
    res = E.evaluate_stream(m, graphs, dev, micro_batch=3, pose_m=pm, pose_s=ps)
    assert np.allclose(res.pred_poses, np.stack(ref_pred), atol=2e-4, rtol=1e-4)
    assert res.t_loss.shape == (7,) and np.isfinite(res.summary()).all()


def test_single_graph_image_streams_agree(dev):
    """A batch that cannot be cut at graph boundaries (one graph: the reference's batch_size=1 loop, test.py:192) has its IMAGES
    spread over model.small_batch_streams HIP streams for the encoder: 1 / 2 / 3 / 4 streams give the same poses (no cross-image
    arithmetic; only the split-K partition of a launch, hence the summation order, depends on how many images it holds: fp32
    within 2e-5, bf16 within one rounding step of the features), for an 8-node and a ragged 5-node graph."""
    import relpose_gnn_amd.synth as S
    m, _ = _build(64, 32, (8, 16, 32, 64), (1, 1, 1, 1), dev)
    for nodes in (8, 5):
        d = _data(S.synth_images(nodes, 32, 40, seed=900 + nodes), nodes, dev)
        for dtype in ("f32", "bf16"):
            m.encoder_dtype = dtype
            outs = []
            for k in (1, 2, 3, 4):
                m.small_batch_streams = k
                a, r, _ = m(d)
                outs.append((a.clone(), r.clone()))
            tol = 2e-5 if dtype == "f32" else 2e-2
            for a, r in outs[1:]:
                assert rel_err(a, outs[0][0]) < tol and rel_err(r, outs[0][1]) < tol, (nodes, dtype)
    m.encoder_dtype, m.small_batch_streams = "f32", 4


def test_eval_stream_input_pipeline_variants_agree(dev):
    """evaluate_stream's input pipeline (pinned double buffers + copy stream, VERDICT r2 missing 3): node images that start in
    pageable host memory, in pinned memory, a mix of both, or on the device must give bit-identical poses -- over ragged graph
    sizes, a ragged last micro-batch and enough micro-batches (6) that both buffers of the pair are reused twice."""
    import relpose_gnn_amd.synth as S
    from relpose_gnn_amd import evaluate as E
    from relpose_gnn_amd.graph import Data, fc_edge_index
    m, _ = _build(64, 32, (8, 16, 32, 64), (1, 1, 1, 1), dev)
    sizes = [8, 4, 8, 6, 8, 8, 5, 8, 8, 8, 3, 8, 8, 8, 8, 7]
    xs = [S.synth_images(n, 32, 40, seed=700 + i) for i, n in enumerate(sizes)]
    ys = [S.hash_normal(f"pipe.y{i}", (n, 6), 0.3) for i, n in enumerate(sizes)]

    def stream(kind):
        out = []
        for i, (x, y) in enumerate(zip(xs, ys)):
            if kind == "pinned" or (kind == "mixed" and i % 3 == 1):
                x = x.clone().pin_memory()
            elif kind == "resident" or (kind == "device_and_host" and i % 2 == 0):
                x = x.to(dev)
            out.append(Data(x=x, edge_index=fc_edge_index(x.shape[0]), y=y))
        return out
    res = {}
    for kind in ("host", "pinned", "mixed", "resident", "device_and_host"):
        st = {}
        res[kind] = E.evaluate_stream(m, stream(kind), dev, micro_batch=3, stats=st).pred_poses
        # staged bytes: everything for the host-resident kinds, nothing for resident; in device_and_host the five mixed
        # micro-batches bypass the pipeline and only the last one (graph 15 alone, host-resident: 7 x 3 x 32 x 40 floats) is staged
        want = {"resident": 0, "device_and_host": sizes[15] * 3 * 32 * 40 * 4}.get(kind, sum(sizes) * 3 * 32 * 40 * 4)
        assert st["micro_batches"] == 6 and st["h2d_bytes"] == want, (kind, st)
        # round 6: the reference's loader delivers PINNED tensors (DataLoader(pin_memory=True), test.py:193) --
        # those must go to the device straight from where they are, with no pageable -> pinned staging copy in between
        per = 3 * 32 * 40 * 4
        want_direct = {"pinned": sum(sizes) * per, "mixed": sum(n for i, n in enumerate(sizes) if i % 3 == 1) * per}.get(kind, 0)
        assert st["direct_bytes"] == want_direct and st["staged_bytes"] == want - want_direct, (kind, st)
    # device_and_host: every micro-batch mixes graphs that live on the device with graphs in host memory (used to
    # die in torch.cat); they are collated on the device, outside the staging pipeline
    for kind in ("pinned", "mixed", "resident", "device_and_host"):
        assert np.array_equal(res[kind], res["host"]), kind
    assert np.isfinite(res["host"]).all() and res["host"].shape == (16, 7)


def test_bf16_encoder_takes_host_rounded_bf16_images(dev):
    """The bf16 encoder rounds its fp32 node images to bf16 first thing (fused stem), so images rounded on the HOST -- what
    evaluate_stream does while it stages them, halving the H2D copy -- must give BIT-identical outputs: forward(x.bfloat16()) ==
    forward(x), and evaluate_stream with bf16 staging == fp32 staging == device-resident fp32 graphs.  The fp32 encoder refuses
    bf16 images; the bf16 encoder takes them with either stem (fused kernel or the three-kernel fallback)."""
    import relpose_gnn_amd.synth as S
    from relpose_gnn_amd import evaluate as E
    from relpose_gnn_amd.graph import Batch, Data, fc_edge_index
    m, _ = _build(64, 32, (64, 16, 32, 64), (1, 1, 1, 1), dev)
    assert not m.accepts_bf16_input
    m.encoder_dtype = "bf16"
    assert m.accepts_bf16_input
    sizes = [8, 4, 8, 6, 8, 8, 5]
    xs = [S.synth_images(n, 32, 40, seed=900 + i) for i, n in enumerate(sizes)]
    ys = [S.hash_normal(f"bfin.y{i}", (n, 6), 0.3) for i, n in enumerate(sizes)]
    graphs = [Data(x=x, edge_index=fc_edge_index(x.shape[0]), y=y) for x, y in zip(xs, ys)]
    b = Batch.from_data_list(graphs[:3]).to(dev)
    a32, r32, _ = m(b)
    b.x = b.x.bfloat16()
    a16, r16, _ = m(b)
    assert torch.equal(a32, a16) and torch.equal(r32, r16)
    res = {}
    for kind, flag in (("bf16_staging", None), ("fp32_staging", False), ("resident", None)):
        st = {}
        gs = graphs if kind != "resident" else [Data(x=g.x.to(dev), edge_index=g.edge_index, y=g.y) for g in graphs]
        res[kind] = E.evaluate_stream(m, gs, dev, micro_batch=3, stats=st, bf16_input=flag).pred_poses
        per_elt = {"bf16_staging": 2, "fp32_staging": 4, "resident": 0}[kind]
        assert st["h2d_bytes"] == per_elt * sum(sizes) * 3 * 32 * 40, (kind, st)
    assert np.array_equal(res["bf16_staging"], res["fp32_staging"]) and np.array_equal(res["resident"], res["fp32_staging"])
    # round 6: fp32 images in PINNED memory (the reference's DataLoader(pin_memory=True), test.py:193).  A rank with few staging
    # threads (the 2-4 of an 8-rank host; here RPG_STAGE_WORKERS = 2) sends them as they are -- no rounding pass, no staging copy --,
    # a rank with >= 8 rounds them on the host (half the H2D bytes: the link, not the forward, bounds the bf16 stream otherwise);
    # bf16_input=True forces the rounding; a stream that mixes pinned and pageable micro-batches uses both pipelines; the poses
    # never change
    pinned = [Data(x=g.x.clone().pin_memory(), edge_index=g.edge_index, y=g.y) for g in graphs]
    total = sum(sizes) * 3 * 32 * 40
    import os
    for kind, gs, flag, want, workers in (("pinned_auto_few_threads", pinned, None, (4 * total, 0, 4 * total), "2"),
                                          ("pinned_auto_many_threads", pinned, None, (2 * total, 2 * total, 0), "16"),
                                          ("pinned_forced_bf16", pinned, True, (2 * total, 2 * total, 0), "2"),
                                          ("pinned_then_pageable", pinned[:3] + graphs[3:], None, None, "2")):
        st = {}
        old_env = os.environ.get("RPG_STAGE_WORKERS")
        os.environ["RPG_STAGE_WORKERS"] = workers
        try:
            got = E.evaluate_stream(m, gs, dev, micro_batch=3, stats=st, bf16_input=flag).pred_poses
        finally:
            if old_env is None:
                del os.environ["RPG_STAGE_WORKERS"]
            else:
                os.environ["RPG_STAGE_WORKERS"] = old_env
        assert np.array_equal(got, res["fp32_staging"]), kind
        if want is not None:
            assert (st["h2d_bytes"], st["staged_bytes"], st["direct_bytes"]) == want, (kind, st)
        else:
            first = sum(sizes[:3]) * 3 * 32 * 40
            assert st["direct_bytes"] == 4 * first and st["staged_bytes"] == 2 * (total - first), (kind, st)
    # RPG_TUNE_FUSED_STEM = 0: the three-kernel stem takes the host-rounded bf16 images too -- a tuning knob must
    # not turn a working evaluation loop into an error -- and agrees with its own fp32-input run bit for bit
    from relpose_gnn_amd import ops
    ops.set_tuning(ops.TUNE_FUSED_STEM, 0)
    try:
        assert m.accepts_bf16_input
        a16u, r16u, _ = m(b)
        b.x = b.x.float()
        a32u, r32u, _ = m(b)
        unfused = E.evaluate_stream(m, graphs, dev, micro_batch=3).pred_poses
    finally:
        ops.set_tuning(ops.TUNE_FUSED_STEM, 1)
    assert torch.equal(a16u, a32u) and torch.equal(r16u, r32u)
    assert rel_err(a16u, a32) < 2e-2 and np.isfinite(unfused).all()           # fused vs three-kernel stem: one extra bf16 rounding
    b.x = b.x.bfloat16()
    m.encoder_dtype = "f32"
    with pytest.raises(TypeError):
        m(b)


def test_eval_harness_with_the_reference_default_knn(dev):
    """The reference's default flags (`--knn 4`, test.py:308): the model rebuilds the graph from the encoder features
    (posenet.py:1047-1048) and eval_RP post-processes the edge list the MODEL returns.  evaluate_stream (micro-batches of 3
    graphs, model-built edge lists cut per graph) == oracle forward with knn=4 on every graph alone + oracle post-processing."""
    import relpose_gnn_amd.synth as S
    from oracle import posenet_ref as O
    from relpose_gnn_amd import evaluate as E
    from relpose_gnn_amd.graph import Data, fc_edge_index
    from relpose_gnn_amd.posenet import PoseNetX_R2
    from relpose_gnn_amd.resnet import ResNet
    planes, blocks, D = (8, 16, 32, 64), (1, 1, 1, 1), 64
    m = PoseNetX_R2(ResNet(blocks, planes), droprate=0.0, pretrained=False, feat_dim=D, edge_feat_dim=D, node_dim=D,
                    input_img_height=32, use_gnn=True, knn=4, use_AP=True, gnn_recursion=2)
    sd = S.synth_state_dict(S.posenet_r2_param_shapes(D, D, D, planes, blocks), seed=1)
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    graphs, ref_pred = [], []
    pm, ps = np.array([0.5, -1.0, 2.0]), np.array([2.0, 3.0, 0.5])
    for i in range(7):
        x = S.synth_images(8, 32, 40, seed=400 + i)
        y = S.hash_normal(f"evalknn.y{i}", (8, 6), 0.3)
        graphs.append(Data(x=x, edge_index=fc_edge_index(8), y=y))
        _, rel, ei = O.posenet_forward(sd, x, fc_edge_index(8), 32, 2, knn=4, batch=torch.zeros(8, dtype=torch.int64))
        assert ei.shape == (2, 32)                                              # 8 nodes x 4 neighbours
        raw = O.query_pose_from_relative(rel.numpy().astype(np.float64), y.numpy().astype(np.float64), ei.numpy())
        ref_pred.append(np.hstack((raw[:3] * ps + pm, O.qexp(raw[3:]))))
    res = E.evaluate_stream(m, graphs, dev, micro_batch=3, pose_m=pm, pose_s=ps)
    assert np.allclose(res.pred_poses, np.stack(ref_pred), atol=2e-4, rtol=1e-4)


def test_multi_stream_equals_single_stream(dev):
    """The batch cut at graph boundaries over 1 / 2 / 3 HIP streams gives the same poses (ragged graph sizes: 8, 4, 8,
    8, 4, 8, 8 nodes) and still flags an edge that leaves its graph."""
    import relpose_gnn_amd.synth as S
    from relpose_gnn_amd.graph import Batch, Data, fc_edge_index
    m, _ = _build(64, 32, (8, 16, 32, 64), (1, 1, 1, 1), dev)
    sizes = [8, 4, 8, 8, 4, 8, 8]
    graphs = [Data(x=S.synth_images(n, 32, 40, seed=200 + i), edge_index=fc_edge_index(n)) for i, n in enumerate(sizes)]
    b = Batch.from_data_list(graphs).to(dev)
    outs = []
    for streams in (1, 2, 3):
        m.hip_streams = streams
        a, r, ei = m(b)
        outs.append((a.cpu(), r.cpu()))
        assert ei is b.edge_index
    for a, r in outs[1:]:
        assert rel_err(a, outs[0][0]) < 1e-5 and rel_err(r, outs[0][1]) < 1e-5
    # an explicit schedule: two groups queued on one stream beside a third group on another (PoseNetX_R2.stream_schedule)
    m.hip_streams = 2
    m.stream_schedule = [(0, 3, 0), (3, 5, 1), (5, 7, 1)]
    a, r, _ = m(b)
    m.stream_schedule = None
    assert rel_err(a.cpu(), outs[0][0]) < 1e-5 and rel_err(r.cpu(), outs[0][1]) < 1e-5
    with pytest.raises(ValueError):
        m.stream_schedule = [(0, 3, 0), (3, 6, 1)]                 # does not cover the batch
        m(b)
    m.stream_schedule = None
    m.hip_streams = 2
    bad = Batch.from_data_list(graphs).to(dev)
    bad.edge_index[0, -1] = 0                      # last graph's edge pointing into the first graph
    m(bad)                                         # counted by the second stream's graph_prepare (slot 1)
    with pytest.raises(IndexError):
        m.check_edge_index()


def test_gnn_node_split_equals_reference_formulation(dev, golden_dir):
    """Per-node precompute of the split Linears (default) vs the reference formulation (3-source gathered GEMMs): same
    poses up to fp32 summation order, at D=64 against golden G4 and at the R3 width against each other."""
    import relpose_gnn_amd.synth as S
    from relpose_gnn_amd import ops
    g = np.load(os.path.join(golden_dir, "g4_full_small.npz"))
    m, _ = _build(64, 32, (8, 16, 32, 64), (1, 1, 1, 1), dev)
    d = _data(S.synth_images(16, 32, 40, seed=3), 8, dev)
    big, _ = _build(2048, 64, (64, 128, 256, 512), (3, 4, 6, 3), dev)
    db = _data(S.synth_images(8 * 6, 64, 64, seed=21), 8, dev)
    outs = {}
    try:
        for split in (1, 0):
            ops.set_tuning(ops.TUNE_GNN_SPLIT, split)
            a, r, _ = m(d)
            assert rel_err(a.cpu(), g["abs"]) < TOL and rel_err(r.cpu(), g["rel"]) < TOL
            outs[split] = tuple(t.cpu() for t in big(db)[:2])
    finally:
        ops.set_tuning(ops.TUNE_GNN_SPLIT, 1)
    ea, er = rel_err(outs[1][0], outs[0][0]), rel_err(outs[1][1], outs[0][1])
    _report("gnn_node_split_vs_reference_formulation_R3_64px", ea, er)
    assert ea < TOL and er < TOL


def test_gnn_fused_aggregation_equals_reference_order(dev):
    """Default GNN forward (attention rows + mean aggregation fused, att.W on node rows, dual-stored ReLU of the edge update)
    vs the reference's order of the same operations (RPG_TUNE_GNN_FUSE_AGG = 0: per-edge attention rows, att.W on edge rows,
    separate scatter-mean): same poses up to fp32 summation order, at the R3 width, both recursions, 3 ragged graphs."""
    import relpose_gnn_amd.synth as S
    from relpose_gnn_amd import ops
    from relpose_gnn_amd.graph import Batch, Data, fc_edge_index
    m, _ = _build(2048, 64, (64, 128, 256, 512), (3, 4, 6, 3), dev)
    graphs = [Data(x=S.synth_images(k, 64, 64, seed=300 + i), edge_index=fc_edge_index(k)) for i, k in enumerate((8, 5, 8))]
    b = Batch.from_data_list(graphs).to(dev)
    m.hip_streams = 1
    a1, r1, _ = m(b)
    ops.set_tuning(ops.TUNE_GNN_FUSE_AGG, 0)
    try:
        a0, r0, _ = m(b)
    finally:
        ops.set_tuning(ops.TUNE_GNN_FUSE_AGG, 1)
    ea, er = rel_err(a1, a0), rel_err(r1, r0)
    _report("gnn_fused_aggregation_vs_reference_order_R3_64px", ea, er)
    assert ea < TOL and er < TOL, (ea, er)
    # the bf16 GNN has TWO implementations as well: the epilogue-emitted bf16 operands aliased onto unused fp32
    # workspace buffers (default, fuse_agg on) and the older convert-per-Linear path (fuse_agg off).  Same bf16 roundings of
    # the same operands up to where the mean is taken, so they agree far inside the bf16 bar (2e-2 on the rel poses).
    m.gnn_dtype = "bf16"
    try:
        ab1, rb1, _ = m(b)
        ops.set_tuning(ops.TUNE_GNN_FUSE_AGG, 0)
        ab0, rb0, _ = m(b)
    finally:
        ops.set_tuning(ops.TUNE_GNN_FUSE_AGG, 1)
        m.gnn_dtype = "f32"
    eab, erb = rel_err(ab1, ab0), rel_err(rb1, rb0)
    _report("gnn_bf16_epilogue_emitted_vs_convert_per_linear_R3_64px", eab, erb)
    # each of the two is within the bf16-GNN bars of the fp32 result (rel 1.5 x 2e-2, abs 1.5 x 5e-2: test_hip_bench_geometry.py),
    # so they are within twice that of each other (measured r4: rel 7.9e-3, abs 7.0e-2 -- the random abs head amplifies ~9x)
    assert erb < 2 * 1.5 * 2e-2 and eab < 2 * 1.5 * 5e-2, (eab, erb)
    assert rel_err(rb1, r1) < 1e-1            # sanity only: bf16 GNN vs fp32 GNN is bounded at the benched geometries (measured here 5e-2)


def test_graph_replay_equals_eager(dev):
    """The forward captured into a HIP graph (both worker streams joined by events) replays to the same poses, also
    after the static input has been overwritten with a new batch."""
    import relpose_gnn_amd.synth as S
    from relpose_gnn_amd.graphed import GraphedForward
    m, _ = _build(64, 32, (8, 16, 32, 64), (1, 1, 1, 1), dev)
    d1 = _data(S.synth_images(8 * 6, 32, 40, seed=31), 8, dev)
    d2 = _data(S.synth_images(8 * 6, 32, 40, seed=32), 8, dev)
    e1 = tuple(t.clone() for t in m(d1)[:2])
    e2 = tuple(t.clone() for t in m(d2)[:2])
    runner = GraphedForward(m, _data(S.synth_images(8 * 6, 32, 40, seed=31), 8, dev))
    a, r, ei = runner(d1)
    assert torch.equal(a, e1[0]) and torch.equal(r, e1[1]) and ei.shape == (2, 56 * 6)
    a, r, _ = runner(d2)
    assert torch.equal(a, e2[0]) and torch.equal(r, e2[1])
    with pytest.raises(ValueError):
        runner(_data(S.synth_images(8 * 4, 32, 40, seed=33), 8, dev))


def test_knn_graph_per_stream_slot_equals_single_stream(dev):
    """Round 5: with ``knn > 0`` (the reference's default CLI, testing/test.py:308) the batch still runs on several streams: every
    slot builds the kNN graph of ITS graphs from its own encoder output (posenet.py:1047-1048).  1 / 2 / 3 streams return the same
    edge list (bit-exact, whole-batch node ids, batch order) and the same poses; graph sizes 8, 4, 8, 8, 4, 8, 8 with k = 3."""
    import relpose_gnn_amd.synth as S
    from relpose_gnn_amd.graph import Batch, Data, fc_edge_index
    m, _ = _build(64, 32, (8, 16, 32, 64), (1, 1, 1, 1), dev)
    sizes = [8, 4, 8, 8, 4, 8, 8]
    graphs = [Data(x=S.synth_images(n, 32, 40, seed=300 + i), edge_index=fc_edge_index(n)) for i, n in enumerate(sizes)]
    b = Batch.from_data_list(graphs).to(dev)
    m.knn = 3
    outs = []
    try:
        for streams in (1, 2, 3):
            m.hip_streams = streams
            a, r, ei = m(b)
            m.check_edge_index()
            assert ei.shape == (2, 3 * sum(sizes)) and r.shape == (3 * sum(sizes), 6)
            outs.append((a.cpu(), r.cpu(), ei.cpu()))
        # an explicit k keeps the single-stream path (posenet.py:1043-1046,1088-1089) and agrees with it
        a_k, r_k, ei_k = m(b, k=3)
        assert torch.equal(ei_k.cpu(), outs[0][2])
    finally:
        m.knn, m.hip_streams = -1, 2
    for a, r, ei in outs[1:]:
        assert torch.equal(ei, outs[0][2])
        assert rel_err(a, outs[0][0]) < 1e-5 and rel_err(r, outs[0][1]) < 1e-5
    # every edge stays inside its graph and points at its query node (row 1), k neighbours each, in node order
    first = np.concatenate([[0], np.cumsum(sizes)])
    e = outs[0][2].numpy()
    assert (np.searchsorted(first, e[0], side="right") == np.searchsorted(first, e[1], side="right")).all()
    assert np.array_equal(e[1], np.repeat(np.arange(sum(sizes)), 3))


def test_dropout_on_the_multi_stream_path_with_seeded_masks(dev):
    """Round 5: ``droprate > 0`` (test.py:309 defaults to 0.5) keeps the multi-stream schedule; dropout + heads run per stream
    slot (posenet.py:1073-1086).  With the generator re-seeded the masks are reproduced in the order the slots draw them --
    (nodes, edges) of slot 0, then of slot 1 -- and the oracle's heads on the identically masked features equal the outputs."""
    import relpose_gnn_amd.synth as S
    from oracle import posenet_ref as O
    m, sd = _build(64, 32, (8, 16, 32, 64), (1, 1, 1, 1), dev, droprate=0.5)
    m.hip_streams = 2
    x = S.synth_images(40, 32, 40, seed=31)                  # 5 graphs: slots of 2 and 3 graphs
    d = _data(x, 8, dev)
    parts = m._partition(d, 40, 5 * 56)
    assert [(p[0], p[1]) for p in parts] == [(0, 16), (16, 40)]
    torch.cuda.manual_seed(77)
    a, r, _ = m(d)
    torch.cuda.synchronize()
    torch.cuda.manual_seed(77)
    mn, me = [], []
    for n0, n1, e0, e1, _ in parts:
        mn.append(torch.nn.functional.dropout(torch.ones(n1 - n0, 64, device=dev), p=0.5))
        me.append(torch.nn.functional.dropout(torch.ones(e1 - e0, 64, device=dev), p=0.5))
    oa, orr, _ = O.posenet_forward(sd, x, d.edge_index.cpu(), 32, 2, node_mask=torch.cat(mn).cpu(), edge_mask=torch.cat(me).cpu())
    ea, er = rel_err(a.cpu(), oa), rel_err(r.cpu(), orr)
    _report("dropout_seeded_masks_two_stream_slots", ea, er)
    assert ea < TOL and er < TOL, (ea, er)


def test_reference_eval_loop_as_written_batch_size_1(dev):
    """The 3-line drop-in of INTEGRATION.md exercised as the reference's caller is written (testing/test.py:205-251): one graph
    per iteration, ``model(data.to(device))``, ``output.size()``, ``output_R.cpu().data.numpy().reshape((-1, s[-1]))``,
    ``len(data)``, ``edge_index.cpu().data.numpy()``, first edge into node 0, target[src] - rel, qexp, un-normalisation --
    against ``evaluate_stream`` (micro-batched, same module) and the oracle forward + oracle post-processing."""
    import relpose_gnn_amd.synth as S
    from oracle import posenet_ref as O
    from relpose_gnn_amd import evaluate as E
    from relpose_gnn_amd.graph import Batch, Data, fc_edge_index
    m, sd = _build(64, 32, (8, 16, 32, 64), (1, 1, 1, 1), dev)
    pose_m, pose_s = np.array([0.5, -1.0, 2.0]), np.array([2.0, 3.0, 0.5])
    graphs = [Data(x=S.synth_images(8, 32, 40, seed=500 + i), edge_index=fc_edge_index(8), y=S.hash_normal(f"lit.y{i}", (8, 6), 0.3),
                   edge_attr=None) for i in range(5)]
    pred_poses, targ_poses, batch_size, ref_node = [], [], 1, 0
    for batch_idx, g in enumerate(graphs):
        data = Batch.from_data_list([g])                     # DataLoader(batch_size=1) collation (test.py:192-194)
        batch_size_ = min(len(data), batch_size)
        output, output_R, edge_index = m(data.to(dev))
        s = output.size()
        output_R = output_R.cpu().data.numpy().reshape((-1, s[-1]))
        target = data.y.to("cpu").numpy().reshape((-1, s[-1]))
        edges = edge_index.cpu().data.numpy()
        ref_idx = np.argwhere(edges[1] == 0)[ref_node, 0]
        out = np.expand_dims(target[edges[0, ref_idx], :] - output_R[ref_idx, :], axis=0)
        out = np.hstack((out[:, :3], np.asarray(tuple(E.qexp(p[3:]) for p in out))))
        target = np.hstack((target[:, :3], np.asarray(tuple(E.qexp(p[3:]) for p in target))))
        out[:, :3] = out[:, :3] * pose_s + pose_m
        target[:, :3] = target[:, :3] * pose_s + pose_m
        for j in range(batch_size_):
            pred_poses.append(out[0])
            targ_poses.append(target[0])
            assert len(pred_poses) == batch_idx * batch_size + j + 1
    pred_poses, targ_poses = np.array(pred_poses), np.array(targ_poses)
    res = E.evaluate_stream(m, graphs, dev, micro_batch=4, pose_m=pose_m, pose_s=pose_s)
    assert np.allclose(pred_poses, res.pred_poses, atol=2e-5) and np.allclose(targ_poses, res.targ_poses, atol=1e-6)
    want = []
    for g in graphs:
        _, rel, _ = O.posenet_forward(sd, g.x, fc_edge_index(8), 32, 2)
        raw = O.query_pose_from_relative(rel.numpy().astype(np.float64), g.y.numpy().astype(np.float64), fc_edge_index(8).numpy())
        want.append(np.hstack((raw[:3] * pose_s + pose_m, O.qexp(raw[3:]))))
    assert np.allclose(pred_poses, np.stack(want), atol=2e-4, rtol=1e-4)


@pytest.mark.parametrize("knn", [-1, 4])
def test_reference_eval_loop_through_lookahead(dev, knn):
    """relpose_gnn_amd.lookahead: the loop of test.py:205-251 literally, once over the plain loader and module
    (one forward per graph) and once over ``lookahead(loader, model, device, micro_batch=4)`` (three forwards for 10 graphs,
    images through the pinned staging pipeline, host tensors back) -- the same predicted and target poses.  knn = 4: the
    module builds the edge list (posenet.py:1047-1048) and the adapter cuts it per graph."""
    import relpose_gnn_amd.synth as S
    from relpose_gnn_amd import evaluate as E
    from relpose_gnn_amd.graph import Batch, Data, fc_edge_index
    from relpose_gnn_amd.lookahead import lookahead
    from relpose_gnn_amd.posenet import PoseNetX_R2
    from relpose_gnn_amd.resnet import ResNet
    planes, blocks, D = (8, 16, 32, 64), (1, 1, 1, 1), 64
    m = PoseNetX_R2(ResNet(blocks, planes), droprate=0.0, pretrained=False, feat_dim=D, edge_feat_dim=D, node_dim=D,
                    input_img_height=32, use_gnn=True, knn=knn, use_AP=True, gnn_recursion=2)
    m.load_state_dict(S.synth_state_dict(S.posenet_r2_param_shapes(D, D, D, planes, blocks), seed=1))
    m = m.to(dev).eval()
    pose_m, pose_s = np.array([0.5, -1.0, 2.0]), np.array([2.0, 3.0, 0.5])
    graphs = [Data(x=S.synth_images(8, 32, 40, seed=600 + i), edge_index=fc_edge_index(8), y=S.hash_normal(f"la.y{i}", (8, 6), 0.3),
                   edge_attr=None) for i in range(10)]

    class Loader:                                            # DataLoader(data_set, batch_size=1, shuffle=False) of test.py:193
        batch_size = 1

        def __len__(self):
            return len(graphs)

        def __iter__(self):
            return (Batch.from_data_list([g]) for g in graphs)

    def reference_loop(loader, model):                       # test.py:205-251
        pred_poses, targ_poses, batch_size, ref_node = [], [], 1, 0
        for batch_idx, data in enumerate(loader):
            batch_size_ = min(len(data), loader.batch_size)
            output, output_R, edge_index = model(data.to(dev))
            s = output.size()
            output_R = output_R.cpu().data.numpy().reshape((-1, s[-1]))
            target = data.y.to("cpu").numpy().reshape((-1, s[-1]))
            edges = edge_index.cpu().data.numpy()
            ref_idx = np.argwhere(edges[1] == 0)[ref_node, 0]
            out = np.expand_dims(target[edges[0, ref_idx], :] - output_R[ref_idx, :], axis=0)
            out = np.hstack((out[:, :3], np.asarray(tuple(E.qexp(p[3:]) for p in out))))
            target = np.hstack((target[:, :3], np.asarray(tuple(E.qexp(p[3:]) for p in target))))
            out[:, :3] = out[:, :3] * pose_s + pose_m
            target[:, :3] = target[:, :3] * pose_s + pose_m
            for j in range(batch_size_):
                pred_poses.append(out[0])
                targ_poses.append(target[0])
                assert len(pred_poses) == batch_idx * batch_size + j + 1
        return np.array(pred_poses), np.array(targ_poses)

    plain_p, plain_t = reference_loop(Loader(), m)
    loader, wrapped = lookahead(Loader(), m, dev, micro_batch=4)
    ahead_p, ahead_t = reference_loop(loader, wrapped)
    assert wrapped.forwards == 3 and wrapped.direct_calls == 0 and ahead_p.shape == (10, 7)
    assert np.allclose(ahead_p, plain_p, atol=2e-5) and np.array_equal(ahead_t, plain_t)
    res = E.evaluate_stream(m, graphs, dev, micro_batch=4, pose_m=pose_m, pose_s=pose_s)
    assert np.allclose(ahead_p, res.pred_poses, atol=5e-6)   # the same micro-batches through the same pipeline


def test_lookahead_direct_calls_are_ordered_and_announced(dev):
    """A call inside the wrapped loop that does not present the graph just yielded (here: data.x cloned) runs the module directly:
    correct results, counted in ``direct_calls``, announced ONCE by a RuntimeWarning (round 6), and ordered against the batched
    forwards in BOTH directions (with one stream both use the module's default workspaces) -- the poses of every
    graph still equal the plain loop's."""
    import warnings
    import relpose_gnn_amd.synth as S
    from relpose_gnn_amd.graph import Batch, Data, fc_edge_index
    from relpose_gnn_amd.lookahead import lookahead
    m, _ = _build(64, 32, (8, 16, 32, 64), (1, 1, 1, 1), dev)
    m.hip_streams = 1
    graphs = [Data(x=S.synth_images(8, 32, 40, seed=900 + i), edge_index=fc_edge_index(8), y=S.hash_normal(f"ld.y{i}", (8, 6), 0.3)) for i in range(9)]

    class Loader:
        batch_size = 1

        def __len__(self):
            return len(graphs)

        def __iter__(self):
            return (Batch.from_data_list([g]) for g in graphs)

    plain = [m(Batch.from_data_list([g]).to(dev))[1].cpu() for g in graphs]
    loader, wrapped = lookahead(Loader(), m, dev, micro_batch=2)
    got = []
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        for i, data in enumerate(loader):
            d = data.to(dev)
            if i % 2 == 1:
                d.x = d.x.clone()                    # not the ticket any more: falls to a direct forward, no synchronisation here
            got.append(wrapped(d)[1])
    got = [g.cpu() for g in got]
    assert wrapped.direct_calls == 4 and wrapped.forwards == 5
    assert sum(issubclass(x.category, RuntimeWarning) and "lookahead" in str(x.message) for x in w) == 1
    for a, b in zip(got, plain):
        assert torch.allclose(a, b, atol=2e-5, rtol=1e-5)


def test_foreign_torchvision_style_feature_extractor(dev):
    """any module with torchvision's ResNet attribute layout works as ``feature_extractor`` -- only its
    ``state_dict()`` (torchvision's key names), ``.avgpool`` and ``.fc.in_features`` are read (posenet.py:942-945).  Here a
    plain nn.Module tree that is NOT relpose_gnn_amd.resnet.ResNet (torchvision itself is absent from this image): the
    constructor replaces its avgpool / fc like the reference does, and the forward equals the package's own ResNet's."""
    import torch.nn as nn
    import relpose_gnn_amd.synth as S
    from relpose_gnn_amd.posenet import PoseNetX_R2

    class Block(nn.Module):                                  # torchvision.models.resnet.BasicBlock's attribute names
        def __init__(self, cin, c, stride):
            super().__init__()
            self.conv1, self.bn1 = nn.Conv2d(cin, c, 3, stride, 1, bias=False), nn.BatchNorm2d(c)
            self.conv2, self.bn2 = nn.Conv2d(c, c, 3, 1, 1, bias=False), nn.BatchNorm2d(c)
            self.downsample = None
            if stride != 1 or cin != c:
                self.downsample = nn.Sequential(nn.Conv2d(cin, c, 1, stride, bias=False), nn.BatchNorm2d(c))

    class Foreign(nn.Module):
        def __init__(self, planes, blocks):
            super().__init__()
            self.conv1, self.bn1 = nn.Conv2d(3, planes[0], 7, 2, 3, bias=False), nn.BatchNorm2d(planes[0])
            self.relu, self.maxpool = nn.ReLU(inplace=True), nn.MaxPool2d(3, 2, 1)
            cin = planes[0]
            for li, (c, nb) in enumerate(zip(planes, blocks), start=1):
                layer = []
                for b in range(nb):
                    layer.append(Block(cin, c, 2 if (li > 1 and b == 0) else 1))
                    cin = c
                setattr(self, f"layer{li}", nn.Sequential(*layer))
            self.avgpool, self.fc = nn.AdaptiveAvgPool2d((1, 1)), nn.Linear(planes[-1], 1000)

    planes, blocks, D = (8, 16, 32, 64), (1, 2, 1, 1), 64
    sd = S.synth_state_dict(S.posenet_r2_param_shapes(D, D, D, planes, blocks), seed=5)
    outs = []
    from relpose_gnn_amd.resnet import ResNet
    for fe in (Foreign(planes, blocks), ResNet(blocks, planes)):
        m = PoseNetX_R2(fe, droprate=0.0, pretrained=False, feat_dim=D, edge_feat_dim=D, node_dim=D, input_img_height=32,
                        use_gnn=True, knn=-1, use_AP=True, gnn_recursion=2)
        assert not isinstance(m.feature_extractor, ResNet) or fe.__class__ is ResNet
        assert m.feature_extractor.fc.out_features == D
        m.load_state_dict(sd)                                 # torchvision key names: strict load
        m = m.to(dev).eval()
        a, r, _ = m(_data(S.synth_images(16, 32, 40, seed=6), 8, dev))
        outs.append((a.cpu(), r.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
