"""World-size-2 test of the graph sharding + pose all-gather on CPU (gloo).  The per-rank "model" is the CPU oracle's
GNN at D=64 so the gathered result can be compared with the unsharded computation."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_graphs, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    import relpose_gnn_amd.synth as S
    from oracle import posenet_ref as O
    from relpose_gnn_amd.shard import gather_rows, shard_counts, shard_range
    sd = S.synth_state_dict(S.posenet_r2_param_shapes(64, 64, 64, (8, 16, 32, 64), (1, 1, 1, 1)), seed=1)
    feats = S.hash_normal("dist.feat", (n_graphs * 8, 64))
    lo, hi = shard_range(n_graphs, rank, world)
    _, rel = O.gnn_forward(sd, feats[lo * 8: hi * 8], O.batch_edge_index(8, hi - lo), 2)
    full = gather_rows(rel.view(hi - lo, 56, 6), shard_counts(n_graphs, world))
    torch.save(full, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_graphs", [4, 5])          # even split and ragged tail
def test_sharded_stream_equals_unsharded(tmp_path, n_graphs):
    import relpose_gnn_amd.synth as S
    from oracle import posenet_ref as O
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_graphs, str(tmp_path)), nprocs=world, join=True)
    sd = S.synth_state_dict(S.posenet_r2_param_shapes(64, 64, 64, (8, 16, 32, 64), (1, 1, 1, 1)), seed=1)
    feats = S.hash_normal("dist.feat", (n_graphs * 8, 64))
    _, rel = O.gnn_forward(sd, feats, O.batch_edge_index(8, n_graphs), 2)
    for r in range(world):
        got = torch.load(os.path.join(str(tmp_path), f"r{r}.pt"))
        assert got.shape == (n_graphs, 56, 6)
        assert torch.allclose(got.view(-1, 6), rel, atol=1e-5, rtol=1e-5)


def test_shard_ranges():
    from relpose_gnn_amd.shard import shard_counts, shard_range
    for n in (1, 7, 8, 2000, 17000):
        for w in (1, 2, 4, 8):
            rs = [shard_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n and all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
            assert max(shard_counts(n, w)) - min(shard_counts(n, w)) <= 1


def _eval_worker(rank, world, port, out_dir, n_graphs=7):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    import numpy as np
    from relpose_gnn_amd import evaluate as E
    from relpose_gnn_amd.graph import Data, fc_edge_index

    class Fake:          # rel = y[dst] - y[src]: deterministic, CPU-only stand-in for the model
        def __call__(self, b):
            return None, b.y[b.edge_index[1]] - b.y[b.edge_index[0]], b.edge_index

    rng = np.random.RandomState(3)
    graphs = [Data(x=torch.zeros(8, 12), edge_index=fc_edge_index(8), y=torch.from_numpy(rng.randn(8, 6) * 0.2).float())
              for _ in range(n_graphs)]
    res = E.evaluate_stream(Fake(), graphs, "cpu", micro_batch=2, rank=rank, world=world)
    np.save(os.path.join(out_dir, f"e{rank}.npy"), res.pred_poses)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_graphs", [(2, 7), (4, 17), (8, 17), (8, 5)])
def test_sharded_evaluation_stream(tmp_path, world, n_graphs):
    """evaluate_stream over 2 / 4 / 8 ranks == the single-process result, on every rank: ragged tails (4 + 3; 17 graphs over
    4 ranks = 5,4,4,4 and over 8 ranks = 3,2,...,2) and ranks WITHOUT a graph (5 graphs on 8 ranks: the zero-row blocks are
    padded for the all-gather and trimmed) -- the world sizes of BASELINE.json configs[3] / [4]."""
    import numpy as np
    from relpose_gnn_amd import evaluate as E
    from relpose_gnn_amd.graph import Data, fc_edge_index
    mp.spawn(_eval_worker, args=(world, _free_port(), str(tmp_path), n_graphs), nprocs=world, join=True)

    class Fake:
        def __call__(self, b):
            return None, b.y[b.edge_index[1]] - b.y[b.edge_index[0]], b.edge_index

    rng = np.random.RandomState(3)
    graphs = [Data(x=torch.zeros(8, 12), edge_index=fc_edge_index(8), y=torch.from_numpy(rng.randn(8, 6) * 0.2).float())
              for _ in range(n_graphs)]
    ref = E.evaluate_stream(Fake(), graphs, "cpu", micro_batch=3).pred_poses
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"e{r}.npy"))
        assert got.shape == (n_graphs, 7) and np.allclose(got, ref, atol=1e-6)


@pytest.mark.parametrize("world", [2, 8])
def test_bench_spawns_its_own_ranks(tmp_path, capsys, world):
    """`python bench.py --gpus N` without a torchrun environment starts N ranks as a child process and relays rank 0's
    JSON line.  The launcher is exercised here with a stand-in rank script on gloo (bench.py's own
    ranks need GPUs)."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("rpg_bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    stub = tmp_path / "rank.py"
    stub.write_text(
        "import json, os, sys\n"
        "import torch, torch.distributed as dist\n"
        "dist.init_process_group('gloo')\n"
        "t = torch.tensor([float(dist.get_rank() + 1)])\n"
        "dist.all_reduce(t)\n"
        "print('noise from rank', dist.get_rank(), flush=True)\n"
        "if dist.get_rank() == 0:\n"
        "    print(json.dumps({'metric': 'stub', 'value': float(t), 'n_gpus': dist.get_world_size(), 'argv': sys.argv[1:]}), flush=True)\n"
        "dist.barrier(); dist.destroy_process_group()\n")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    rc = bench.spawn_ranks(world, script=str(stub), argv=["--gpus", str(world), "--steps", "3"])
    out = capsys.readouterr().out.strip().splitlines()
    assert rc == 0 and len(out) == 1                        # exactly ONE line on stdout: the JSON
    line = json.loads(out[0])
    assert line["n_gpus"] == world and line["value"] == world * (world + 1) / 2 and line["argv"] == ["--gpus", str(world), "--steps", "3"]


def _bind_child(q, local_rank, local_world):
    from relpose_gnn_amd import shard
    before = sorted(os.sched_getaffinity(0))
    got = shard.bind_rank_to_host_slice(local_rank, local_world)
    again = shard.bind_rank_to_host_slice(local_rank, local_world)       # idempotent: the narrowed mask is not cut again
    q.put((before, got, again, sorted(os.sched_getaffinity(0))))


def test_rank_host_slices_partition_the_cpus():
    """Host placement of the ranks (round 5): equal contiguous CPU slices, every CPU in exactly one; the sysfs cpulist parser;
    and -- in a child process, so that pytest keeps its own mask -- bind_rank_to_host_slice narrows the affinity once."""
    from relpose_gnn_amd import shard
    assert shard._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    cpus = list(range(128))
    for w in (1, 2, 4, 8):
        parts = [shard.rank_cpu_slice(r, w, cpus) for r in range(w)]
        assert sorted(c for p in parts for c in p) == cpus and max(map(len, parts)) - min(map(len, parts)) <= 1
        if w == 8:
            assert all(max(p) < 64 for p in parts[:4]) and all(min(p) >= 64 for p in parts[4:])     # socket-major numbering
    assert shard.rank_cpu_slice(5, 8, [0, 1, 2]) == [2]                  # more ranks than CPUs: still one CPU each
    if not hasattr(os, "sched_setaffinity") or len(os.sched_getaffinity(0)) < 2:
        pytest.skip("no affinity control here")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_bind_child, args=(q, 1, 2))
    p.start()
    before, got, again, after = q.get(timeout=120)
    p.join(30)
    assert got == sorted(shard._slice_in_order(1, 2, shard._core_major(before)))          # whole cores, SMT siblings together
    assert again == got and after == got and sorted(os.sched_getaffinity(0)) == before


def _report_worker(rank, world, port, out_dir, share_gpu):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import json
    from relpose_gnn_amd.shard import rank_report
    # rank r "took" (10 + r) ms per step over 4 steps; rank 1 is given 3 CPUs, the others 2; with share_gpu ranks 0 and 1 claim
    # the same device identity (a launcher that forgot LOCAL_RANK)
    ident = 1000 + (0 if (share_gpu and rank == 1) else rank)
    rep = rank_report("cpu", 4 * (10 + rank) * 1e-3, steps=4, host_cpus=3 if rank == 1 else 2, identity=ident)
    with open(os.path.join(out_dir, f"rep{rank}.json"), "w") as f:
        json.dump(rep, f)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,share_gpu", [(2, False), (8, False), (8, True)])
def test_rank_report_diagnoses_a_multi_rank_run(tmp_path, world, share_gpu):
    """Round 6: the fields a `--gpus N` line carries come from collectives, not from the launcher's
    configuration -- ranks that took part (all_reduce of ones), distinct devices (all_gather of each rank's device identity:
    two ranks on one GPU show as distinct_gpus < ranks), per-rank step time min / max / which rank, CPUs per rank.  Same dict
    on every rank."""
    import json
    mp.spawn(_report_worker, args=(world, _free_port(), str(tmp_path), share_gpu), nprocs=world, join=True)
    reps = [json.load(open(os.path.join(str(tmp_path), f"rep{r}.json"))) for r in range(world)]
    assert all(r == reps[0] for r in reps[1:])
    r = reps[0]
    assert r["rccl_ranks_seen"] == world
    assert r["distinct_gpus"] == (world - 1 if share_gpu else world) and r["distinct_hosts"] == 1
    assert r["rank_ms_min"] == 10.0 and r["rank_ms_max"] == 10.0 + world - 1 and r["fastest_rank"] == 0 and r["slowest_rank"] == world - 1
    assert abs(r["rank_ms_spread"] - (world - 1) / 10.0) < 1e-3 and abs(r["rank_ms_mean"] - (10.0 + (world - 1) / 2)) < 1e-3
    assert r["host_cpus_min"] == 2 and r["host_cpus_max"] == 3


def test_rank_report_without_a_process_group():
    from relpose_gnn_amd.shard import rank_report
    r = rank_report("cpu", 0.05, steps=5, host_cpus=7)
    assert r["rccl_ranks_seen"] is None and r["rank_ms_max"] == 10.0 and r["distinct_gpus"] == 1 and r["host_cpus_min"] == 7


def test_rank_host_slice_restores_the_callers_mask():
    """a library call (evaluate_stream) must not narrow its caller's CPU mask for good -- shard.rank_host_slice binds
    for the duration of a block and puts the previous mask back; a process its entry script bound is left alone."""
    import multiprocessing as pmp
    ctx = pmp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_slice_child, args=(q,))
    p.start()
    before, inside, after, inside2, after2 = q.get(timeout=60)
    p.join(timeout=30)
    if len(before) < 2:
        pytest.skip("one CPU: nothing to narrow")
    assert inside is not None and set(inside) < set(before) and after == before       # narrowed inside, restored after
    assert inside2 == inside and after2 == sorted(inside)                              # bound by the "entry script": left alone both ways


def _slice_child(q):
    from relpose_gnn_amd import shard
    before = sorted(os.sched_getaffinity(0))
    with shard.rank_host_slice(0, 2) as cpus:
        inside = sorted(os.sched_getaffinity(0))
        assert cpus is None or sorted(cpus) == inside
    after = sorted(os.sched_getaffinity(0))
    shard.bind_rank_to_host_slice(0, 2)                     # process-long (what bench.py / tools/eval_stream.py do)
    with shard.rank_host_slice(0, 2):
        inside2 = sorted(os.sched_getaffinity(0))
    q.put((before, inside, after, inside2, sorted(os.sched_getaffinity(0))))
