#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/ by running the REFERENCE's own model code.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden.py

The reference's ``PoseNetX_R2`` / ``simpleConvEdge_upt`` / ``AttentionBlock`` classes are imported
unmodified from /root/reference/python.  Their third-party imports that are not installed here
(torch_geometric, torch_cluster, transforms3d) are satisfied by the minimal stand-ins defined
below; the only stand-in that carries semantics is ``MessagePassing.propagate`` (PyG 2.0.1:
``*_j`` arguments gather at edge_index[0], ``*_i`` at edge_index[1], mean aggregation at
edge_index[1] = sum / max(count,1), then ``update``).  torchvision's resnet34 is replaced by
``oracle/resnet_module.py``.

For every case the script (1) checks that the functional oracle ``oracle/posenet_ref.py`` agrees
with the reference output, and (2) writes the *reference* outputs as fixtures.  Weights and inputs
come from the repo's deterministic hash generator (relpose-gnn_amd/synth.py) and are therefore not
stored.  Nothing here travels as source of the reference: the fixtures are numbers only.
"""
import inspect
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/python"


# --------------------------------------------------------------------------- #
# stand-ins for the un-installed third-party modules
# --------------------------------------------------------------------------- #
class MessagePassing(torch.nn.Module):
    def __init__(self, aggr="add", flow="source_to_target", **kw):
        super().__init__()
        assert flow == "source_to_target"
        self.aggr = aggr

    def propagate(self, edge_index, size=None, **kw):
        n = size[1] if size is not None else int(edge_index.max()) + 1
        margs = {}
        for name in inspect.signature(self.message).parameters:
            if name.endswith("_j"):
                margs[name] = kw[name[:-2]].index_select(0, edge_index[0])
            elif name.endswith("_i"):
                margs[name] = kw[name[:-2]].index_select(0, edge_index[1])
            else:
                margs[name] = kw[name]
        msg = self.message(**margs)
        out = torch.zeros(n, msg.shape[1], dtype=msg.dtype).index_add_(0, edge_index[1], msg)
        if self.aggr == "mean":
            cnt = torch.zeros(n, dtype=msg.dtype).index_add_(0, edge_index[1], torch.ones(msg.shape[0]))
            out = out / cnt.clamp(min=1).unsqueeze(1)
        elif self.aggr != "add":
            raise NotImplementedError(self.aggr)
        uargs = {k: kw[k] for k in inspect.signature(self.update).parameters if k in kw}
        return self.update(out, **uargs)

    def message(self, x_j):
        return x_j

    def update(self, aggr_out):
        return aggr_out


def _knn_graph(x, k, batch=None, loop=False, **kw):
    """torch_cluster is not installed: the reference's kNN call sites (posenet.py:1044-1050) get the oracle's restatement
    of the published algorithm.  This pins the reference's control flow around the kNN graph (which edge list feeds the
    GNN, which one is returned), NOT the neighbour search itself."""
    from oracle import posenet_ref as O
    assert not loop
    return O.knn_graph(x, k, batch)


def install_stubs():
    tg = types.ModuleType("torch_geometric")
    tgnn = types.ModuleType("torch_geometric.nn")
    tgconv = types.ModuleType("torch_geometric.nn.conv")
    tc = types.ModuleType("torch_cluster")
    tgnn.knn_graph = _knn_graph
    tc.knn_graph = _knn_graph
    tgconv.MessagePassing = MessagePassing
    tg.nn, tgnn.conv = tgnn, tgconv
    t3 = types.ModuleType("transforms3d")
    t3.euler = types.ModuleType("transforms3d.euler")
    t3.quaternions = types.ModuleType("transforms3d.quaternions")
    for name, mod in (("torch_geometric", tg), ("torch_geometric.nn", tgnn), ("torch_geometric.nn.conv", tgconv),
                      ("torch_cluster", tc), ("transforms3d", t3), ("transforms3d.euler", t3.euler),
                      ("transforms3d.quaternions", t3.quaternions)):
        sys.modules[name] = mod
    sys.path.insert(0, REF)


def rel_err(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def main():
    install_stubs()
    from niantic.modules.posenet import PoseNetX_R2            # the reference model
    from niantic.utils import pose_utils as ref_pu              # reference qexp / angular error
    from oracle import posenet_ref as O
    from oracle.resnet_module import ResNetCPU
    import relpose_gnn_amd.synth as S
    from relpose_gnn_amd.graph import fc_edge_index

    torch.manual_seed(0)
    torch.set_num_threads(8)
    out = {}

    # ---- G1: FC edge lists --------------------------------------------------------------
    n4 = [[0, 1, 2, 0, 1, 0, 1, 2, 3, 2, 3, 3], [1, 2, 3, 2, 3, 3, 0, 1, 2, 0, 1, 0]]
    assert O.fc_edge_index(4).tolist() == n4 and fc_edge_index(4).tolist() == n4
    assert torch.equal(O.fc_edge_index(8), fc_edge_index(8))
    e8 = O.fc_edge_index(8)
    assert e8.shape == (2, 56) and e8[:, 28].tolist() == [1, 0]
    np.savez(os.path.join(HERE, "g1_fc_edges.npz"), n4=O.fc_edge_index(4).numpy(), n8=e8.numpy())

    def build_ref(D, img_h, planes, blocks):
        fe = ResNetCPU(blocks=blocks, planes=planes)
        m = PoseNetX_R2(fe, droprate=0.0, pretrained=False, feat_dim=D, edge_feat_dim=D, node_dim=D,
                        input_img_height=img_h, use_gnn=True, knn=-1, use_AP=True, gnn_recursion=2)
        shapes = S.posenet_r2_param_shapes(D, D, D, planes, blocks)
        ref_sd = m.state_dict()
        assert list(ref_sd.keys()) == list(shapes.keys()), "state-dict key order/name mismatch"
        for k, v in ref_sd.items():
            assert tuple(v.shape) == tuple(shapes[k]), (k, v.shape, shapes[k])
        sd = S.synth_state_dict(shapes, seed=1)
        m.load_state_dict(sd)
        return m.eval(), sd

    # ---- inventory check at the R3 dims (2048) against the instantiated reference ---------
    fe = ResNetCPU()
    m_full = PoseNetX_R2(fe, droprate=0.0, pretrained=False, feat_dim=2048, edge_feat_dim=2048, node_dim=2048,
                         input_img_height=224, use_gnn=True, knn=-1, use_AP=True, gnn_recursion=2)
    shapes_full = S.posenet_r2_param_shapes()
    ref_sd = m_full.state_dict()
    assert list(ref_sd.keys()) == list(shapes_full.keys())
    assert all(tuple(ref_sd[k].shape) == tuple(shapes_full[k]) for k in shapes_full)
    n_param = sum(p.numel() for p in m_full.parameters())
    print("R3 param count", n_param, "tensors", len(ref_sd))
    assert n_param == 74805836 and len(ref_sd) == 248

    # ---- G2/G3: GNN-only at D=64 (B=1 and B=3) ------------------------------------------
    small_planes, small_blocks = (8, 16, 32, 64), (1, 1, 1, 1)
    m64, sd64 = build_ref(64, 32, small_planes, small_blocks)

    class _Identity(torch.nn.Module):           # feed node features straight into the GNN
        def forward(self, x):
            return x.view(x.shape[0], -1)[:, :64]

    for tag, B in (("g2", 1), ("g3", 3)):
        feats = S.hash_normal(f"{tag}.feat", (8 * B, 64), 1.0, 0.0, seed=2)
        ei = O.batch_edge_index(8, B)
        stages = {}
        o_abs, o_rel = O.gnn_forward(sd64, feats, ei, 2, stages)
        saved_fe = m64.feature_extractor
        m64.feature_extractor = _Identity()
        m64.input_img_height = 1
        with torch.no_grad():
            r_abs, r_rel, r_ei = m64(types.SimpleNamespace(x=torch.cat([feats, torch.zeros(8 * B, 128)], 1),
                                                           edge_index=ei, edge_attr=None, batch=None))
        m64.feature_extractor = saved_fe
        m64.input_img_height = 32
        print(tag, "oracle vs reference rel err abs/rel:", rel_err(o_abs, r_abs), rel_err(o_rel, r_rel))
        assert rel_err(o_abs, r_abs) < 2e-6 and rel_err(o_rel, r_rel) < 2e-6 and torch.equal(r_ei, ei)
        np.savez(os.path.join(HERE, f"{tag}_gnn_d64_b{B}.npz"), abs=r_abs.numpy(), rel=r_rel.numpy(),
                 **{"stage_" + k: v.numpy() for k, v in stages.items()})

    # ---- G4: full model, small encoder (planes 8..64, one block per layer), 32x40 input ----
    x = S.synth_images(8 * 2, 32, 40, seed=3)
    ei = O.batch_edge_index(8, 2)
    stages = {}
    o_abs, o_rel, _ = O.posenet_forward(sd64, x, ei, 32, 2, stages)
    with torch.no_grad():
        r_abs, r_rel, _ = m64(types.SimpleNamespace(x=x, edge_index=ei, edge_attr=None, batch=None))
        r_feat = m64.feature_extractor(x.view(16, 3, 32, -1))
    print("g4 oracle vs reference:", rel_err(o_abs, r_abs), rel_err(o_rel, r_rel), rel_err(stages["fc"], r_feat))
    assert rel_err(o_abs, r_abs) < 2e-6 and rel_err(o_rel, r_rel) < 2e-6 and rel_err(stages["fc"], r_feat) < 2e-6
    np.savez(os.path.join(HERE, "g4_full_small.npz"), abs=r_abs.numpy(), rel=r_rel.numpy(), feat=r_feat.numpy())

    # ---- G4b: real ResNet34 layout at 64x64, D=64, randomised BN, per-stage checks -------
    m34, sd34 = build_ref(64, 64, (64, 128, 256, 512), (3, 4, 6, 3))
    x = S.synth_images(4, 64, 64, seed=4)
    ei = O.fc_edge_index(4)
    stages = {}
    o_abs, o_rel, _ = O.posenet_forward(sd34, x, ei, 64, 2, stages)
    with torch.no_grad():
        r_abs, r_rel, _ = m34(types.SimpleNamespace(x=x, edge_index=ei, edge_attr=None, batch=None))
        fe = m34.feature_extractor
        t = fe.relu(fe.bn1(fe.conv1(x.view(4, 3, 64, 64))))
        ref_st = {"stem": t}
        t = fe.maxpool(t)
        for li in range(1, 5):
            t = getattr(fe, f"layer{li}")(t)
            ref_st[f"layer{li}"] = t
        ref_st["fc"] = fe.fc(torch.flatten(fe.avgpool(t), 1))
    for k, v in ref_st.items():
        assert rel_err(stages[k], v) < 2e-6, k
    print("g4b oracle vs reference:", rel_err(o_abs, r_abs), rel_err(o_rel, r_rel),
          {k: float(v.abs().max()) for k, v in ref_st.items()})
    assert rel_err(o_abs, r_abs) < 2e-6 and rel_err(o_rel, r_rel) < 2e-6
    np.savez(os.path.join(HERE, "g4b_resnet34_64px.npz"), abs=r_abs.numpy(), rel=r_rel.numpy(),
             feat=ref_st["fc"].numpy(),
             **{"l2_" + k: np.float64(v.double().norm().item()) for k, v in ref_st.items()})

    # ---- G5: full size (D=2048, 224x224), one 4-node graph: outputs + stage norms ---------
    sd_full = S.synth_state_dict(shapes_full, seed=1)
    m_full.load_state_dict(sd_full)
    m_full.eval()
    x = S.synth_images(4, 224, 224, seed=5)
    stages = {}
    o_abs, o_rel, _ = O.posenet_forward(sd_full, x, ei, 224, 2, stages)
    with torch.no_grad():
        r_abs, r_rel, _ = m_full(types.SimpleNamespace(x=x, edge_index=ei, edge_attr=None, batch=None))
    print("g5 oracle vs reference:", rel_err(o_abs, r_abs), rel_err(o_rel, r_rel),
          "| max|feat|", float(stages["fc"].abs().max()), "max|abs|", float(r_abs.abs().max()))
    assert rel_err(o_abs, r_abs) < 1e-5 and rel_err(o_rel, r_rel) < 1e-5
    np.savez(os.path.join(HERE, "g5_full_r3_224.npz"), abs=r_abs.numpy(), rel=r_rel.numpy(),
             **{"l2_" + k: np.float64(v.double().norm().item()) for k, v in stages.items()})

    # ---- G7: constructor flags use_attention=True, use_AP=False, L=2 (extra unused gnn2 weights) ----------------
    fe = ResNetCPU(blocks=small_blocks, planes=small_planes)
    m7 = PoseNetX_R2(fe, droprate=0.0, pretrained=False, feat_dim=64, edge_feat_dim=64, node_dim=64, input_img_height=32,
                     use_gnn=True, use_attention=True, knn=-1, use_AP=False, gnn_recursion=2, L=2)
    shapes7 = S.posenet_r2_param_shapes(64, 64, 64, small_planes, small_blocks, use_attention=True, use_AP=False, L=2)
    assert list(m7.state_dict().keys()) == list(shapes7.keys())
    assert all(tuple(v.shape) == tuple(shapes7[k]) for k, v in m7.state_dict().items())
    sd7 = S.synth_state_dict(shapes7, seed=7)
    m7.load_state_dict(sd7)
    m7.eval()
    x = S.synth_images(16, 32, 40, seed=3)
    ei = O.batch_edge_index(8, 2)
    o_abs, o_rel, _ = O.posenet_forward(sd7, x, ei, 32, 2, use_attention=True, use_AP=False)
    with torch.no_grad():
        r_abs, r_rel, _ = m7(types.SimpleNamespace(x=x, edge_index=ei, edge_attr=None, batch=None))
    print("g7 oracle vs reference:", rel_err(o_abs, r_abs), rel_err(o_rel, r_rel), tuple(r_abs.shape))
    assert r_abs.shape == (112, 6) and rel_err(o_abs, r_abs) < 2e-6 and rel_err(o_rel, r_rel) < 2e-6
    np.savez(os.path.join(HERE, "g7_flags_att_noAP_L2.npz"), abs=r_abs.numpy(), rel=r_rel.numpy())

    # ---- G8: kNN control flow (knn=3 in the constructor; forward(data, k=2)) with the stand-in neighbour search -----
    x = S.synth_images(16, 32, 40, seed=9)
    ei = O.batch_edge_index(8, 2)
    bvec = torch.arange(2).repeat_interleave(8)
    m64.knn = 3
    with torch.no_grad():
        a_c, r_c, e_c = m64(types.SimpleNamespace(x=x, edge_index=ei, edge_attr=None, batch=bvec))
        a_k, r_k, e_k = m64(types.SimpleNamespace(x=x, edge_index=ei, edge_attr=None, batch=bvec), k=2)
    m64.knn = -1
    with torch.no_grad():
        a_f, r_f, e_f = m64(types.SimpleNamespace(x=x, edge_index=ei, edge_attr=None, batch=bvec), k=2)
    o1 = O.posenet_forward(sd64, x, ei, 32, 2, knn=3, batch=bvec)
    o2 = O.posenet_forward(sd64, x, ei, 32, 2, knn=3, k=2, batch=bvec)
    o3 = O.posenet_forward(sd64, x, ei, 32, 2, k=2, batch=bvec)
    for (ra, rr, re), (oa, orr, oe) in (((a_c, r_c, e_c), o1), ((a_k, r_k, e_k), o2), ((a_f, r_f, e_f), o3)):
        assert torch.equal(re, oe) and rel_err(oa, ra) < 2e-6 and rel_err(orr, rr) < 2e-6
    assert e_c.shape == (2, 48) and e_k.shape == (2, 32) and r_k.shape == (48, 6) and r_f.shape == (32, 6)
    print("g8 kNN control flow: oracle == reference (with stand-in knn_graph)")
    np.savez(os.path.join(HERE, "g8_knn_flow.npz"), abs_ctor=a_c.numpy(), rel_ctor=r_c.numpy(), ei_ctor=e_c.numpy(),
             abs_both=a_k.numpy(), rel_both=r_k.numpy(), ei_both=e_k.numpy(),
             abs_fwd=a_f.numpy(), rel_fwd=r_f.numpy(), ei_fwd=e_f.numpy())

    # ---- G9: on-disk artefacts written by the REFERENCE's own code -------------------------------
    # (a) a checkpoint through niantic.utils.utils.save_checkpoint (utils.py:22-31), the G4 model (D=64, planes 8..64),
    #     the criterion and the Adam optimiser built as training/train.py:196-214 builds them, after ONE optimiser step with
    #     lr=0 / weight_decay=0 (weights bit-unchanged; the file then carries a real Adam state: step counters, exp_avg ...).
    # (b) a graph sample in the layout of dataset_7Scenes_multi.py:437-446 (`torch.save(Data(x, edge_index, y, edge_attr))`):
    #     torch_geometric is not installed, so the Data / GlobalStorage classes are stand-ins that reproduce PyG 2.0.1's
    #     attribute layout and class paths (Data.__dict__['_store'] -> GlobalStorage.__dict__['_mapping']); the numbers are
    #     graph 0 of the G4 input.  tests: file -> relpose_gnn_amd.io -> load_state_dict -> forward == G4.
    tv = types.ModuleType("torchvision")
    tv.datasets = types.ModuleType("torchvision.datasets")
    tv.datasets.folder = types.ModuleType("torchvision.datasets.folder")
    tv.datasets.folder.default_loader = lambda path: None
    for name, mod in (("torchvision", tv), ("torchvision.datasets", tv.datasets), ("torchvision.datasets.folder", tv.datasets.folder)):
        sys.modules.setdefault(name, mod)
    from niantic.utils.utils import save_checkpoint                # the reference's writer
    from niantic.modules.criterion import PoseNetCriterion
    m64.load_state_dict(sd64)
    crit = PoseNetCriterion(sax=0.0, saq=-3.0, learn_beta=True)
    crit_R = PoseNetCriterion(sax=0.0, saq=-3.0, learn_beta=True)
    opt = torch.optim.Adam([{"params": m64.parameters()}, {"params": [crit.sax, crit.saq]},
                            {"params": [crit_R.sax, crit_R.saq]}], lr=0.0, weight_decay=0.0)
    x4 = S.synth_images(16, 32, 40, seed=3)
    m64.train()
    a_t, r_t, _ = m64(types.SimpleNamespace(x=x4, edge_index=O.batch_edge_index(8, 2), edge_attr=None, batch=None))
    (a_t.square().mean() + r_t.square().mean()).backward()
    opt.step()
    m64.eval()
    assert all(torch.equal(v, sd64[k]) for k, v in m64.state_dict().items() if "num_batches_tracked" not in k and "running_" not in k)
    m64.load_state_dict(sd64)                                       # (train-mode BN moved the running statistics: put them back)
    save_checkpoint(HERE, 199, m64, opt, crit)
    ck = torch.load(os.path.join(HERE, "epoch_199.pth.tar"), weights_only=False)
    assert set(ck) == {"epoch", "model_state_dict", "optim_state_dict", "criterion_state_dict"} and ck["epoch"] == 199
    assert all(torch.equal(ck["model_state_dict"][k], sd64[k]) for k in sd64) and len(ck["optim_state_dict"]["state"]) >= 60

    pyg = {n: types.ModuleType(n) for n in ("torch_geometric.data", "torch_geometric.data.data", "torch_geometric.data.storage")}

    class GlobalStorage:
        def __init__(self, parent):
            self._mapping, self._parent = {}, parent                # PyG 2.0.1 BaseStorage.__getstate__ stores the parent itself

    class Data:
        def __init__(self, **kw):
            self._store = GlobalStorage(self)
            self._store._mapping.update(kw)
    GlobalStorage.__module__, GlobalStorage.__qualname__ = "torch_geometric.data.storage", "GlobalStorage"
    Data.__module__, Data.__qualname__ = "torch_geometric.data.data", "Data"
    pyg["torch_geometric.data.data"].Data, pyg["torch_geometric.data.storage"].GlobalStorage = Data, GlobalStorage
    sys.modules.update(pyg)
    ei8 = O.fc_edge_index(8)
    y8 = S.hash_normal("g9.y", (8, 6), 0.3, 0.0, seed=9)
    os.makedirs(os.path.join(HERE, "processed"), exist_ok=True)
    torch.save(Data(x=x4[:8].view(8, -1).clone(), edge_index=ei8, y=y8, edge_attr=y8[ei8[1]] - y8[ei8[0]]),        # y[target] - y[source], dataset_7Scenes_multi.py:425-429
               os.path.join(HERE, "processed", "data_000000.pt"))
    for n in pyg:
        sys.modules.pop(n, None)
    print("g9 checkpoint (reference save_checkpoint) + graph sample written:",
          os.path.getsize(os.path.join(HERE, "epoch_199.pth.tar")), os.path.getsize(os.path.join(HERE, "processed", "data_000000.pt")), "bytes")

    # ---- G6: caller-side pose utilities ------------------------------------------------------
    rng = np.random.RandomState(7)
    v = rng.randn(6, 3) * 0.7
    v[0] = 0.0
    q = np.stack([ref_pu.qexp(r) for r in v])
    ang = np.array([ref_pu.quaternion_angular_error(q[i], q[(i + 1) % 6]) for i in range(6)])
    for i in range(6):
        assert np.allclose(O.qexp(v[i]), q[i])
        assert np.isclose(O.quaternion_angular_error(q[i], q[(i + 1) % 6]), ang[i])
    np.savez(os.path.join(HERE, "g6_pose_utils.npz"), v=v, q=q, ang=ang)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
