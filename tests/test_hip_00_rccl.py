"""RCCL on the real GPU at world size 1 ("What's missing" 1): every distributed test elsewhere is gloo on CPU,
so until this file nothing had initialised RCCL.  The box has one GPU, so the world is one rank -- but it is the same
code the 2/4/8-GPU runs execute: ``init_process_group("nccl")``, the all-gather of ``shard.gather_rows`` on device
tensors, ``barrier`` / ``all_reduce``, and ``bench.py`` through its own ``torch.distributed.run`` command line
(``spawn_ranks``).  The reference is single-process (/root/reference/python/niantic/testing/test.py:78-80): there is
nothing to compare with but the unsharded result.

Everything runs in CHILD processes (a rank is its own process; this file is named to be collected before the other
GPU tests, while the pytest process has not touched the GPU yet)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    # dmabuf IPC is the only flavour the pool's host driver supports (bench.spawn_ranks explains); never override a choice
    return dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


_CHILD = r"""
import os, sys
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
from relpose_gnn_amd.shard import gather_rows, shard_counts, shard_range
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
rel = torch.randn(5, 56, 6, device=dev)
out = gather_rows(rel, shard_counts(5, 1), always=True)        # the collective itself: all_gather_into_tensor over RCCL
assert out.data_ptr() != rel.data_ptr() and torch.equal(out, rel)
assert gather_rows(rel, [5]) is rel                            # default: a single rank skips the collective
t = torch.tensor([3.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 3.5
from relpose_gnn_amd.shard import rank_report
rep = rank_report(dev, 0.02, steps=2)                          # all_reduce of ones + all_gather of the device identities over RCCL
assert rep["rccl_ranks_seen"] == 1 and rep["distinct_gpus"] == 1 and rep["rank_ms_max"] == 10.0 and rep["slowest_rank"] == 0, rep
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL_OK", torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else "")
"""


def test_rccl_world1_gather_rows_on_device():
    r = subprocess.run([sys.executable, "-c", _CHILD.format(root=ROOT, port=_free_port())], capture_output=True, text=True,
                       timeout=600, env=_env())
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])


def test_bench_through_its_own_launcher_world1():
    """`python bench.py --gpus 1 --launcher`: spawn_ranks' torch.distributed.run command line -> rank 0 initialises RCCL,
    runs the per-step all-gather of the relative poses, the barriers and the all_reduce(MAX) of the elapsed time."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--launcher", "--steps", "2", "--warmup", "1",
           "--graphs", "4", "--cpu-baseline-seconds", "0", "--no-kernel-timing"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=_env(), cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.lstrip().startswith("{") and '"metric"' in ln]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["value"] > 0 and rec["config"]["process_group"].startswith("nccl (RCCL), world_size=1")
    cfg = rec["config"]                                      # round 6: what the collectives observed, as flat scalars
    assert rec["rccl_ranks_seen"] == 1 and cfg["distinct_gpus"] == 1 and cfg["slowest_rank"] == 0 and cfg["allgather_ms"] > 0
    assert abs(cfg["rank_ms_max"] - rec["ms_per_step"]) / rec["ms_per_step"] < 0.2 and isinstance(cfg["loaded_library"], str)
    assert all(not isinstance(v, (dict,)) for v in cfg.values())


def test_eval_stream_tool_under_its_launcher_world1():
    """tools/eval_stream.py (the configs[3] / [4] multi-rank driver) under its own ``torch.distributed.run`` command line at
    world size 1: rank 0 initialises RCCL, evaluate_stream shards the stream (one block), the [G, 14] pose rows go through the
    real all_gather_into_tensor, then barrier + all_reduce(MAX) of the elapsed time."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tools", "eval_stream.py"), "--graphs", "24", "--shape", "64x96",
           "--micro-batch", "8", "--pool", "8"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=_env(), cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.lstrip().startswith("{") and '"graphs_per_s"' in ln]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["rccl_ranks_seen"] == 1 and rec["graphs"] == 24 and rec["graphs_per_s"] > 0
    assert rec["distinct_gpus"] == 1 and rec["slowest_rank"] == 0 and rec["rank_ms_max"] > 0 and rec["staged_gb"] > 0 and rec["direct_gb"] == 0
