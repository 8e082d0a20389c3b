"""Multi-GPU sharding of independent query graphs (one process per GPU, torch.distributed; backend "nccl" is RCCL).

The reference is single-process / single-GPU (/root/reference/python/niantic/testing/test.py:78-80,192); every graph of
an evaluation stream is independent (PyG batching keeps the node-id blocks disjoint), so the stream shards
embarrassingly: rank r takes a contiguous block of graphs, runs the same replicated weights, and the only exchange is
one all-gather of the predicted relative poses (56 x 6 floats = 1,344 bytes per 8-node graph) so that every rank
(rank 0 for the evaluation bookkeeping of test.py:213-251) holds the whole stream's result.  The payload is KBs:
latency-bound on xGMI, no bucketing needed.
"""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.distributed as dist


def shard_range(n_graphs: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of graph ids for ``rank``; the first ``n_graphs % world`` ranks get one extra."""
    q, r = divmod(n_graphs, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_counts(n_graphs: int, world: int) -> List[int]:
    return [shard_range(n_graphs, r, world)[1] - shard_range(n_graphs, r, world)[0] for r in range(world)]


def gather_rows(local: torch.Tensor, counts: List[int], group=None, always: bool = False) -> torch.Tensor:
    """All-gather per-rank row blocks of unequal length: ``local`` is [counts[rank], ...]; returns the concatenation
    [sum(counts), ...] in rank order on every rank.  Ragged tails are zero-padded to max(counts) for the collective
    (one ``all_gather_into_tensor``) and trimmed afterwards.  A single rank returns ``local`` itself unless ``always`` is
    set and a process group exists: then the (RCCL) collective runs at world size 1 too, which is how the one-GPU box
    exercises the multi-GPU step (bench.py under its own launcher, tests/test_hip_rccl.py)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and not (always and dist.is_initialized()):
        return local
    rank = dist.get_rank(group)
    assert local.shape[0] == counts[rank], (local.shape, counts, rank)
    m = max(counts)
    pad = local
    if local.shape[0] < m:
        pad = torch.zeros((m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
    out = torch.empty((world * m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
    if all(c == m for c in counts):
        return out
    return torch.cat([out[r * m: r * m + c] for r, c in enumerate(counts)], 0)
