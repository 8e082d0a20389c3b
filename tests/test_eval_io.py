"""Host-side rows next to the hot path: evaluation post-processing and the PyG-free readers."""
import os
import sys
import types

import numpy as np
import pytest
import torch

from oracle import posenet_ref as O
from relpose_gnn_amd import evaluate as E
from relpose_gnn_amd.graph import Data, fc_edge_index


def test_pose_utils_against_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "g6_pose_utils.npz"))       # written by the reference's pose_utils
    for i in range(6):
        assert np.allclose(E.qexp(g["v"][i]), g["q"][i], atol=1e-12)
        assert np.isclose(E.quaternion_angular_error(g["q"][i], g["q"][(i + 1) % 6]), g["ang"][i], atol=1e-9)


def test_query_pose_rule():
    rng = np.random.RandomState(0)
    ei = fc_edge_index(8).numpy()
    rel, y = rng.randn(56, 6) * 0.1, rng.randn(8, 6) * 0.3
    assert E.reference_edge(ei) == 28 and ei[:, 28].tolist() == [1, 0]
    m, s = np.array([1.0, 2.0, 3.0]), np.array([2.0, 2.0, 0.5])
    pred, targ = E.query_pose(rel, y, ei, m, s)
    raw = O.query_pose_from_relative(rel, y, ei)                       # oracle statement of test.py:227-232
    assert np.allclose(pred[:3], raw[:3] * s + m) and np.allclose(pred[3:], O.qexp(raw[3:]))
    assert np.allclose(targ[:3], y[0, :3] * s + m) and np.allclose(targ[3:], O.qexp(y[0, 3:]))
    res = E.errors(np.stack([pred, targ]), np.stack([targ, targ]))
    assert res.t_loss[1] == 0 and res.q_loss[1] < 1e-5 and res.t_loss[0] > 0
    with pytest.raises(ValueError):
        E.reference_edge(np.array([[0, 1], [1, 2]]))


class _FakeModel:
    """rel = y[dst] - y[src] + 0.01: lets the harness be checked on CPU without the HIP module."""

    def __call__(self, batch):
        ei = batch.edge_index
        return None, batch.y[ei[1]] - batch.y[ei[0]] + 0.01, ei


def test_evaluate_stream_and_npz(tmp_path):
    rng = np.random.RandomState(1)
    graphs = []
    for _ in range(5):
        y = torch.from_numpy(rng.randn(8, 6) * 0.2).float()
        graphs.append(Data(x=torch.zeros(8, 12), edge_index=fc_edge_index(8), y=y))
    res = E.evaluate_stream(_FakeModel(), graphs, "cpu", micro_batch=2)
    assert res.pred_poses.shape == (5, 7)
    for i, g in enumerate(graphs):                                      # pred = y[1] - (y[0]-y[1]+0.01)
        y = g.y.numpy().astype(np.float64)
        exp = y[1] - (y[0] - y[1] + 0.01)
        assert np.allclose(res.pred_poses[i, :3], exp[:3], atol=1e-6)
        assert np.allclose(res.pred_poses[i, 3:], E.qexp(exp[3:]), atol=1e-6)
    assert len(res.summary()) == 4 and res.median_t == float(np.median(res.t_loss))
    # 7-Scenes rel_path bookkeeping of test.py:255-260: data_<linear id>.pt -> the linear id-th colour frame of the split
    frames = [f"/data/7scenes/chess/seq-03/frame-{i:06d}.color.png" for i in range(12)]
    files = [f"/graphs/chess_fc8_sp5_test/processed/data_{i:06d}.pt" for i in (0, 3, 4, 7, 11)]
    rel = E.seven_scenes_rel_paths(files, frames, "/data/7scenes")
    assert rel[1] == "chess/seq-03/frame-000003.color.png" and rel[4].endswith("frame-000011.color.png")
    E.save_poses(tmp_path / "out.npz", res, rel)
    z = np.load(tmp_path / "out.npz")
    assert set(z.files) == {"rel_path", "abs_t", "abs_q", "targ_t", "targ_q"} and z["abs_q"].shape == (5, 4)
    assert list(z["rel_path"]) == rel
    with pytest.raises(AssertionError):
        E.save_poses(tmp_path / "bad.npz", res, rel[:-1])                # the reference asserts equal lengths (test.py:40)


class _FakeKnnModel:
    """Returns its OWN edge list (every node's single nearest-by-index neighbour ring, grouped by target like
    torch_cluster.knn_graph), as the reference model does with knn > 0 (posenet.py:1047-1048,1088-1089)."""

    def __call__(self, batch):
        n = batch.x.shape[0]
        tgt = torch.arange(n).repeat_interleave(2)
        base = (tgt // 8) * 8
        src = base + (tgt - base + torch.tensor([1, 3]).repeat(n)) % 8
        ei = torch.stack([src, tgt])
        return None, batch.y[ei[1]] - batch.y[ei[0]] + 0.01, ei


def test_evaluate_stream_with_model_built_edges():
    """The reference's default --knn 4 (test.py:308): eval_RP post-processes the edge list the MODEL returns."""
    rng = np.random.RandomState(2)
    graphs = [Data(x=torch.zeros(8, 12), edge_index=fc_edge_index(8), y=torch.from_numpy(rng.randn(8, 6) * 0.2).float())
              for _ in range(5)]
    res = E.evaluate_stream(_FakeKnnModel(), graphs, "cpu", micro_batch=2)
    for i, g in enumerate(graphs):                     # first edge into node 0 is (1 -> 0): pred = y[1] - (y[0]-y[1]+0.01)
        y = g.y.numpy().astype(np.float64)
        exp = y[1] - (y[0] - y[1] + 0.01)
        assert np.allclose(res.pred_poses[i, :3], exp[:3], atol=1e-6)
        assert np.allclose(res.pred_poses[i, 3:], E.qexp(exp[3:]), atol=1e-6)


def _fake_pyg(layout):
    """Install throw-away modules shaped like PyG so that torch.save writes pickles with torch_geometric class paths."""
    mods = {n: types.ModuleType(n) for n in ("torch_geometric", "torch_geometric.data", "torch_geometric.data.data",
                                             "torch_geometric.data.storage")}

    class GlobalStorage:
        def __init__(self):
            self._mapping = {}
    GlobalStorage.__module__ = "torch_geometric.data.storage"

    class PygData:
        def __init__(self, **kw):
            if layout == "v2":
                self._store = GlobalStorage()
                self._store._mapping.update(kw)
            else:
                self.__dict__.update(kw)
    PygData.__module__, PygData.__qualname__, PygData.__name__ = "torch_geometric.data.data", "Data", "Data"
    GlobalStorage.__qualname__ = "GlobalStorage"
    mods["torch_geometric.data.data"].Data = PygData
    mods["torch_geometric.data.storage"].GlobalStorage = GlobalStorage
    return mods, PygData


@pytest.mark.parametrize("layout", ["v1", "v2"])
def test_graph_reader_without_pyg(tmp_path, layout):
    from relpose_gnn_amd import io as rio
    mods, PygData = _fake_pyg(layout)
    x, ei = torch.randn(8, 30), fc_edge_index(8)
    y = torch.randn(8, 6)
    sample = PygData(x=x, edge_index=ei, y=y, edge_attr=y[ei[1]] - y[ei[0]])
    os.makedirs(tmp_path / "processed")
    sys.modules.update(mods)
    try:
        for i in (0, 1, 10, 2):
            torch.save(sample, tmp_path / "processed" / f"data_{i}.pt")
    finally:
        for n in mods:
            sys.modules.pop(n, None)
    files = rio.processed_files(str(tmp_path))
    # self.file_list.sort() of the reference is lexicographic (dataset_Cambridge_multi.py:72): data_10 before data_2
    assert [os.path.basename(f) for f in files] == ["data_0.pt", "data_1.pt", "data_10.pt", "data_2.pt"]
    d = rio.load_graph(files[3])
    assert torch.equal(d.x, x) and torch.equal(d.edge_index, ei) and torch.equal(d.y, y) and d.edge_attr.shape == (56, 6)
    assert "torch_geometric" not in sys.modules
    bad = tmp_path / "bad.pt"
    torch.save({"z": 1}, bad)
    with pytest.raises(ValueError):
        rio.load_graph(str(bad))
    evil = tmp_path / "evil.pt"
    torch.save({"x": x, "edge_index": ei, "f": os.system}, evil)       # a callable from an arbitrary module
    with pytest.raises(Exception) as exc:
        rio.load_graph(str(evil))
    assert "refusing to unpickle" in str(exc.value)

    class _Rce:                                                         # a callable UNDER the torch root
        def __reduce__(self):
            import torch.utils.collect_env as ce
            return (ce.run, ("echo PWNED > " + str(tmp_path / "pwned"),))
    torch.save({"x": x, "edge_index": ei, "f": _Rce()}, evil)
    with pytest.raises(Exception) as exc:
        rio.load_graph(str(evil))
    assert "refusing to unpickle torch.utils.collect_env.run" in str(exc.value) and not (tmp_path / "pwned").exists()

    class _Nested:                                                      # torch.storage._load_from_bytes unpickles its
        def __reduce__(self):                                           # argument with the DEFAULT module -> nested gadget
            import pickle
            import torch.storage as ts

            class _Inner:
                def __reduce__(self_inner):
                    return (os.system, ("echo PWNED > " + str(tmp_path / "pwned2"),))
            return (ts._load_from_bytes, (pickle.dumps(_Inner()),))
    for reader in (rio.load_graph, rio.load_checkpoint_state_dict):
        torch.save({"x": x, "edge_index": ei, "f": _Nested()}, evil)
        with pytest.raises(Exception):
            reader(str(evil))
        assert not (tmp_path / "pwned2").exists()
    # the legitimate use of that helper (a storage / tensor pickled on its own inside the file) still loads
    import pickle
    import torch.storage as ts

    class _Legit:
        def __reduce__(self):
            import io
            buf = io.BytesIO()
            torch.save(torch.arange(5.0), buf)
            return (ts._load_from_bytes, (buf.getvalue(),))
    torch.save({"x": x, "edge_index": ei, "t": _Legit()}, evil)
    assert torch.equal(rio.load_checkpoint_state_dict(str(evil))["t"], torch.arange(5.0))


def test_checkpoint_reader(tmp_path):
    from relpose_gnn_amd import io as rio
    sd = {"proj_edge.weight": torch.randn(4, 8)}
    torch.save({"epoch": 199, "model_state_dict": sd, "optim_state_dict": {}, "criterion_state_dict": {}}, tmp_path / "epoch_199.pth.tar")
    out = rio.load_checkpoint_state_dict(str(tmp_path / "epoch_199.pth.tar"))
    assert torch.equal(out["proj_edge.weight"], sd["proj_edge.weight"])


def test_reference_written_checkpoint_and_sample(golden_dir):
    """VERDICT r3 item 5(b), CPU half: ``tests/golden/epoch_199.pth.tar`` was written by the REFERENCE's own
    ``save_checkpoint`` (utils/utils.py:22-31; make_golden.py G9: the D=64 model of golden G4 + criterion + an Adam state
    after one step) and ``processed/data_000000.pt`` carries graph 0 of the G4 input in PyG 2.0.1's ``Data`` pickle layout.
    Both go through the PyG-free allow-list readers; the oracle forward on what they return reproduces G4 (graph 0)."""
    import relpose_gnn_amd.synth as S
    from relpose_gnn_amd import io as rio
    sd = rio.load_checkpoint_state_dict(os.path.join(golden_dir, "epoch_199.pth.tar"))
    want = S.synth_state_dict(S.posenet_r2_param_shapes(64, 64, 64, (8, 16, 32, 64), (1, 1, 1, 1)), seed=1)
    assert list(sd.keys()) == list(want.keys()) and all(torch.equal(sd[k], want[k]) for k in want)
    files = rio.processed_files(golden_dir)
    assert [os.path.basename(f) for f in files] == ["data_000000.pt"]
    d = rio.load_graph(files[0])
    assert "torch_geometric" not in sys.modules
    assert torch.equal(d.x, S.synth_images(16, 32, 40, seed=3)[:8]) and torch.equal(d.edge_index, fc_edge_index(8))
    assert d.y.shape == (8, 6) and torch.equal(d.edge_attr, d.y[d.edge_index[1]] - d.y[d.edge_index[0]])
    g = np.load(os.path.join(golden_dir, "g4_full_small.npz"))
    a, r, _ = O.posenet_forward(sd, d.x, d.edge_index, 32, 2)
    assert float((a - torch.from_numpy(g["abs"][:8])).abs().max()) < 1e-5 * float(np.abs(g["abs"]).max())
    assert float((r - torch.from_numpy(g["rel"][:56])).abs().max()) < 1e-5 * float(np.abs(g["rel"]).max())


def test_staging_thread_budget_is_shared_by_the_ranks_of_a_host():
    """Round 5: the host's copy rate peaks at ~16 threads in all (profiles/r5_stage_scale_*.jsonl), so a rank's input pipeline gets
    16 // local_world staging threads (at least 2; one rank alone: 8 for fp32 staging, 16 for bf16), capped by its affinity mask;
    RPG_STAGE_WORKERS overrides."""
    from relpose_gnn_amd.evaluate import staging_workers as w
    assert [w(True, k, 256, "0") for k in (1, 2, 4, 8)] == [8, 8, 4, 2]
    assert [w(False, k, 256, "0") for k in (1, 2, 4, 8, 16)] == [16, 8, 4, 2, 2]
    assert w(False, 1, 6, "0") == 6 and w(True, 8, 1, "0") == 1            # never more than the CPUs it may run on
    assert w(False, 8, 256, "12") == 12 and w(False, 8, 4, "12") == 4      # explicit override, still capped


def test_partition_for_model_built_graphs_needs_only_the_node_table():
    """Round 5: with ``knn > 0`` the model replaces ``data.edge_index``, so the multi-stream cut needs the nodes per graph only
    (PoseNetX_R2._graph_sizes / _partition with e_total=None) -- from this package's Batch, a PyG collation table, ``ptr`` or
    ``batch``; host logic, no GPU."""
    from types import SimpleNamespace
    from relpose_gnn_amd.graph import Batch, Data, fc_edge_index
    from relpose_gnn_amd.posenet import PoseNetX_R2
    sizes = [8, 4, 8, 8, 4]
    b = Batch.from_data_list([Data(x=torch.zeros(n, 12), edge_index=fc_edge_index(n)) for n in sizes])
    gs = PoseNetX_R2._graph_sizes
    assert gs(b, 32, None) == (sizes, [0] * 5) and gs(b, 31, None) is None
    pyg = SimpleNamespace(x=b.x, edge_index=b.edge_index, batch=b.batch, _slice_dict={"x": torch.tensor([0, 8, 12, 20, 28, 32])})
    assert gs(pyg, 32, None) == (sizes, [0] * 5)
    assert gs(SimpleNamespace(x=b.x, edge_index=b.edge_index, batch=b.batch, ptr=torch.tensor([0, 8, 12, 20, 28, 32])), 32, None)[0] == sizes
    assert gs(SimpleNamespace(x=b.x, edge_index=b.edge_index, batch=b.batch), 32, None)[0] == sizes
    fake = SimpleNamespace(hip_streams=2, stream_schedule=None, _graph_sizes=gs)
    parts = PoseNetX_R2._partition(fake, b, 32, None)
    assert [(p[0], p[1], p[4]) for p in parts] == [(0, 12, 0), (12, 32, 1)]
    fake.hip_streams = 3                                                     # 5 graphs < 2 x 3: one stream
    assert PoseNetX_R2._partition(fake, b, 32, None) is None


class _StubLoader:
    """DataLoader(batch_size=1) stand-in: yields single-graph batches, has __len__ / batch_size / dataset (test.py:193,205-207)."""
    batch_size = 1

    def __init__(self, graphs):
        from relpose_gnn_amd.graph import Batch
        self.dataset, self._batch, self.reads = graphs, Batch, 0

    def __len__(self):
        return len(self.dataset)

    def __iter__(self):
        for g in self.dataset:
            self.reads += 1
            yield self._batch.from_data_list([g])


class _StubModel:
    """(abs, rel, edge_index) as a per-node function of the images: any batching gives the same rows."""
    index_check = "deferred"

    def __init__(self, knn=False):
        self.calls, self.knn = [], knn

    def __call__(self, data, k=None):
        f = data.x.reshape(data.x.shape[0], -1).float().mean(1, keepdim=True)
        ab = f * torch.arange(1, 7, dtype=torch.float32)
        ei = data.edge_index
        if self.knn:                                         # a model-built edge list: every stored edge reversed, new tensor
            ei = torch.stack([data.edge_index[1], data.edge_index[0]])
        self.calls.append(int(data.x.shape[0]))
        return ab, ab[ei[1]] - ab[ei[0]], ei


@pytest.mark.parametrize("knn", [False, True])
def test_lookahead_keeps_the_reference_loop_and_batches_the_forwards(knn):
    """relpose_gnn_amd.lookahead: the loop of test.py:205-251 unchanged (enumerate(loader), len(loader), loader.batch_size,
    len(data), model(data.to(device)), .cpu().data.numpy()) over 11 graphs of 8 / 5 nodes with micro_batch = 4 issues three
    forwards (4 + 4 + 3 graphs), reads the loader one micro-batch ahead, and gives every iteration the rows the module returns
    for that graph alone."""
    from relpose_gnn_amd.lookahead import lookahead
    g = torch.Generator().manual_seed(3)
    graphs = [Data(x=torch.randn(n, 3 * 6 * 8, generator=g), edge_index=fc_edge_index(n), y=torch.randn(n, 6, generator=g), edge_attr=None)
              for n in (8, 5, 8, 8, 5, 8, 8, 8, 5, 8, 8)]
    base, model = _StubLoader(graphs), _StubModel(knn)
    loader, wrapped = lookahead(base, model, "cpu", micro_batch=4)
    assert len(loader) == 11 and loader.batch_size == 1 and loader.dataset is graphs and wrapped.index_check == "deferred"
    alone = _StubModel(knn)
    seen = 0
    for batch_idx, data in enumerate(loader):
        assert batch_idx == seen and len(data) == 1
        if batch_idx == 0:
            assert base.reads == 8                            # two micro-batches have been read before the first graph comes out
        output, output_R, edge_index = wrapped(data.to("cpu"))
        s = output.size()
        want = alone(_StubLoader([graphs[batch_idx]])._batch.from_data_list([graphs[batch_idx]]))
        assert s == want[0].size() and torch.equal(data.y, graphs[batch_idx].y) and torch.equal(data.x, graphs[batch_idx].x)
        assert np.array_equal(output.cpu().data.numpy(), want[0].numpy())
        assert np.array_equal(output_R.cpu().data.numpy().reshape((-1, s[-1])), want[1].numpy())
        assert np.array_equal(edge_index.cpu().data.numpy(), want[2].numpy())
        seen += 1
    assert seen == 11 and model.calls == [29, 29, 21] and wrapped.forwards == 3 and wrapped.direct_calls == 0
    # anything that is not the graph just handed out goes to the module itself
    extra = _StubLoader(graphs)._batch.from_data_list(graphs[:2])
    out = wrapped(extra)
    assert wrapped.direct_calls == 1 and model.calls[-1] == 13 and out[0].shape == (13, 6)
    wrapped.index_check = "sync"                              # attribute writes reach the module
    assert model.index_check == "sync"
    with pytest.raises(ValueError):                           # a loader with more than one graph per item is not the reference's loop
        list(lookahead([extra], model, "cpu")[0])
