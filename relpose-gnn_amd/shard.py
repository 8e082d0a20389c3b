"""Multi-GPU sharding of independent query graphs (one process per GPU, torch.distributed; backend "nccl" is RCCL).

The reference is single-process / single-GPU (/root/reference/python/niantic/testing/test.py:78-80,192); every graph of
an evaluation stream is independent (PyG batching keeps the node-id blocks disjoint), so the stream shards
embarrassingly: rank r takes a contiguous block of graphs, runs the same replicated weights, and the only exchange is
one all-gather of the predicted relative poses (56 x 6 floats = 1,344 bytes per 8-node graph) so that every rank
(rank 0 for the evaluation bookkeeping of test.py:213-251) holds the whole stream's result.  The payload is KBs:
latency-bound on xGMI, no bucketing needed.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

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


# ---- host placement of a rank (round 5) -------------------------------------------------------------------------------
# The reference is one process (testing/test.py:78-80,192-194); eight ranks on a two-socket GPU host are eight processes whose
# staging threads (evaluate._InputPipeline: 8-16 per rank) and pinned buffers would otherwise float over both sockets: a
# rank's pageable -> pinned copy and the DMA out of the pinned buffer then cross the socket link for half of the ranks.

_BOUND = None        # ((local_rank, local_world), cpus) once bind_rank_to_host_slice has narrowed this process's mask


def _parse_cpulist(text: str) -> List[int]:
    out: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def gpu_numa_node(device_index: int) -> Optional[int]:
    """NUMA node of GPU ``device_index`` per sysfs (None: unknown / -1)."""
    try:
        p = torch.cuda.get_device_properties(device_index)
        dom, bus, devn = getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", None), getattr(p, "pci_device_id", None)
        if bus is None or devn is None:
            return None
        with open(f"/sys/bus/pci/devices/{dom:04x}:{bus:02x}:{devn:02x}.0/numa_node") as f:
            node = int(f.read().strip())
        return node if node >= 0 else None
    except Exception:
        return None


def gpu_local_cpus(device_index: int) -> Optional[List[int]]:
    """CPUs of the NUMA node GPU ``device_index`` hangs off, from sysfs (PCI address -> local_cpulist), or None if the
    platform does not say (no sysfs entry, numa_node = -1, virtualised PCI topology).  Never touches the GPU runtime beyond
    the cached device properties."""
    try:
        p = torch.cuda.get_device_properties(device_index)
        dom, bus, devn = getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", None), getattr(p, "pci_device_id", None)
        if bus is None or devn is None:
            return None
        base = f"/sys/bus/pci/devices/{dom:04x}:{bus:02x}:{devn:02x}.0"
        with open(os.path.join(base, "numa_node")) as f:
            if int(f.read().strip()) < 0:
                return None
        with open(os.path.join(base, "local_cpulist")) as f:
            cpus = _parse_cpulist(f.read())
        return cpus or None
    except Exception:
        return None


def _core_major(cpus: Sequence[int]) -> List[int]:
    """``cpus`` ordered core by core (a core's SMT siblings next to each other), from sysfs; plain ascending order if the
    topology files are not there.  On the GPU hosts of this pool (2 x 64 cores, SMT on) CPU c and c + 128 are siblings: cut in
    plain numeric order, rank 0 of a NUMA node would get cores 0-31 and rank 2 their hyper-thread twins."""
    def key(c):
        try:
            base = f"/sys/devices/system/cpu/cpu{c}/topology/"
            with open(base + "physical_package_id") as f:
                pkg = int(f.read())
            with open(base + "core_id") as f:
                core = int(f.read())
            return (pkg, core, c)
        except Exception:
            return (0, c, c)
    keys = {c: key(c) for c in cpus}
    return sorted(cpus, key=lambda c: keys[c])


def rank_cpu_slice(local_rank: int, local_world: int, cpus: Optional[Sequence[int]] = None) -> List[int]:
    """The contiguous slice of ``cpus`` (default: this process's current affinity mask, in ascending order) that
    ``local_rank`` of ``local_world`` ranks on this host gets: equal shares, every CPU in exactly one share.  On a
    two-socket host with the usual socket-major CPU numbering ranks 0..W/2-1 land on socket 0 and the rest on socket 1,
    like the GPUs they drive."""
    cpus = sorted(cpus if cpus is not None else os.sched_getaffinity(0))
    if local_world <= 1 or not cpus:
        return list(cpus)
    lo, hi = shard_range(len(cpus), local_rank, local_world)
    return list(cpus[lo:hi]) if hi > lo else [cpus[local_rank % len(cpus)]]


def _slice_in_order(local_rank: int, local_world: int, ordered: Sequence[int]) -> List[int]:
    if local_world <= 1 or not ordered:
        return list(ordered)
    lo, hi = shard_range(len(ordered), local_rank, local_world)
    return list(ordered[lo:hi]) if hi > lo else [ordered[local_rank % len(ordered)]]


def bind_rank_to_host_slice(local_rank: int, local_world: int, device_index: Optional[int] = None) -> Optional[List[int]]:
    """Pin the calling THREAD and every thread started afterwards (``os.sched_setaffinity(0, ...)`` binds the caller, not the
    threads that already exist: the HIP / RCCL runtime threads and torch's intra-op pool keep their masks; staging pools created
    afterwards inherit the new one, and pinned buffers allocated afterwards are first touched from these CPUs) to its share of the host: the CPUs of its GPU's NUMA node
    where sysfs names them -- cut among the ranks whose GPUs share that node -- else an equal contiguous slice of the
    current mask.  Returns the CPU list, or None if nothing was changed (single rank, RPG_BIND_RANKS=0, unsupported
    platform).  Plain ``os.sched_setaffinity``: no exec, no numactl hop (safe after the GPU has been initialised).
    Process-long when an entry script calls it (bench.py, tools/eval_stream.py: they know LOCAL_RANK); a LIBRARY call uses the
    ``rank_host_slice`` context manager below, which puts the previous mask back."""
    global _BOUND
    if local_world <= 1 or os.environ.get("RPG_BIND_RANKS", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    if _BOUND is not None:                 # once per process: a second call would cut the already narrowed mask again
        return _BOUND[1] if _BOUND[0] == (local_rank, local_world) else None
    try:
        allowed = sorted(os.sched_getaffinity(0))
        cpus = None
        dev = local_rank if device_index is None else device_index
        near = gpu_local_cpus(dev)
        if near:
            near = [c for c in near if c in set(allowed)]
            # this rank's place among the ranks whose GPUs hang off the same NUMA node (rank r drives GPU r of this host; where
            # a GPU's node is unknown the ranks are assumed spread evenly: GPUs 0..W/2-1 on node 0 ...)
            my_node = gpu_numa_node(dev)
            nodes_of = [gpu_numa_node(r) for r in range(local_world)] if (my_node is not None and dev == local_rank) else []
            if nodes_of and all(n is not None for n in nodes_of):
                same = [r for r, n in enumerate(nodes_of) if n == my_node]
                per_node, place = len(same), same.index(local_rank)
            else:
                nodes = max(1, round(len(allowed) / max(1, len(near))))
                per_node = max(1, local_world // nodes)
                place = local_rank % per_node
            if near:
                cpus = sorted(_slice_in_order(place, per_node, _core_major(near)))
        if not cpus:
            cpus = sorted(_slice_in_order(local_rank, local_world, _core_major(allowed)))
        os.sched_setaffinity(0, cpus)
        _BOUND = ((local_rank, local_world), cpus)
        return cpus
    except Exception:
        return None


class rank_host_slice:
    """``with rank_host_slice(local_rank, local_world, device_index):`` -- bind_rank_to_host_slice for the duration of the block
    (a library call must not narrow its caller's CPU mask for good).  The threads started inside (the staging pool of
    ``evaluate._InputPipeline``) keep the slice; the calling thread gets its previous mask back on exit.  A process its entry
    script has already bound (``_BOUND`` set before the block) is left alone both ways."""

    def __init__(self, local_rank: int, local_world: int, device_index: Optional[int] = None):
        self.args = (local_rank, local_world, device_index)
        self.prev = None
        self.cpus = None

    def __enter__(self):
        global _BOUND
        if _BOUND is None and hasattr(os, "sched_getaffinity"):
            prev = os.sched_getaffinity(0)
            self.cpus = bind_rank_to_host_slice(*self.args)
            if self.cpus is not None:
                self.prev = prev
        elif _BOUND is not None:
            self.cpus = _BOUND[1]
        return self.cpus

    def __exit__(self, *exc):
        global _BOUND
        if self.prev is not None:
            try:
                os.sched_setaffinity(0, self.prev)
            finally:
                _BOUND = None
        return False


# ---- a multi-rank record that diagnoses itself (round 6) ----------------------------------------------------------------
# The reference is one process (testing/test.py:78-80): nothing in it says which of eight ranks was slow.  Every line a
# multi-rank run prints (bench.py, tools/eval_stream.py) carries what RCCL itself observed -- not what the launcher configured.

def gpu_identity(device) -> int:
    """A 63-bit integer naming the physical GPU behind ``device``: PCI domain / bus / device where the runtime reports them,
    else a hash of the UUID or of the device name + index.  Two ranks that (by a launcher mistake) drive the SAME GPU report
    the same number."""
    d = torch.device(device)
    if d.type != "cuda":
        return (os.getpid() << 8) | 0xff           # CPU stand-in (gloo tests): one "device" per process
    p = torch.cuda.get_device_properties(d)
    bus, devn = getattr(p, "pci_bus_id", None), getattr(p, "pci_device_id", None)
    if bus is not None and devn is not None:
        return (int(getattr(p, "pci_domain_id", 0)) << 16) | (int(bus) << 8) | int(devn)
    import hashlib
    key = str(getattr(p, "uuid", "")) or f"{p.name}#{d.index}"
    return int.from_bytes(hashlib.sha256(key.encode()).digest()[:8], "big") >> 1


def rank_report(device, elapsed_s: float, steps: int = 1, host_cpus: Optional[int] = None, identity: Optional[int] = None,
                group=None) -> dict:
    """Flat scalars describing the run AS THE COLLECTIVES SAW IT (every rank must call; every rank gets the same dict):

      rccl_ranks_seen      all_reduce(SUM) of a one per rank -- the ranks that really took part in a collective
      distinct_gpus        number of different gpu_identity() values among them (== ranks_seen unless ranks share a GPU)
      distinct_hosts       number of different host names (hashed)
      rank_ms_min / _max / _mean, slowest_rank, fastest_rank      per-rank ms per step from an all_gather of each rank's own clock
      rank_ms_spread       (max - min) / min: one slow rank (a throttling GPU, a far socket) shows here, a slow collective does not
      host_cpus_min / _max CPUs in each rank's affinity mask (bind_rank_to_host_slice), min / max over ranks
      numa_nodes_seen      distinct NUMA nodes of the ranks' GPUs (-1 entries = unknown are counted once)

    Works on any backend (the tensors live on ``device`` for nccl, on the CPU for gloo)."""
    d = torch.device(device)
    if not (dist.is_available() and dist.is_initialized()):
        ms = 1e3 * elapsed_s / max(1, steps)
        return {"rccl_ranks_seen": None, "distinct_gpus": 1, "distinct_hosts": 1, "rank_ms_min": round(ms, 3), "rank_ms_max": round(ms, 3),
                "rank_ms_mean": round(ms, 3), "slowest_rank": 0, "fastest_rank": 0, "rank_ms_spread": 0.0,
                "host_cpus_min": host_cpus, "host_cpus_max": host_cpus, "numa_nodes_seen": 1}
    world = dist.get_world_size(group)
    tdev = d if dist.get_backend(group) == "nccl" else torch.device("cpu")
    ones = torch.ones(1, dtype=torch.int64, device=tdev)
    dist.all_reduce(ones, op=dist.ReduceOp.SUM, group=group)
    import hashlib
    import socket
    host_hash = int.from_bytes(hashlib.sha256(socket.gethostname().encode()).digest()[:7], "big")
    ident = gpu_identity(d) if identity is None else int(identity)
    numa = gpu_numa_node(d.index) if d.type == "cuda" else None
    if host_cpus is None and hasattr(os, "sched_getaffinity"):
        host_cpus = len(os.sched_getaffinity(0))
    mine = torch.tensor([ident, host_hash, int(round(1e6 * elapsed_s / max(1, steps))), int(host_cpus or 0), -1 if numa is None else int(numa)],
                        dtype=torch.int64, device=tdev)
    allr = torch.empty(world * mine.numel(), dtype=torch.int64, device=tdev)
    dist.all_gather_into_tensor(allr, mine, group=group)
    rows = allr.view(world, mine.numel()).cpu().tolist()
    ms = [r[2] / 1e3 for r in rows]
    lo, hi = min(ms), max(ms)
    return {"rccl_ranks_seen": int(ones.item()), "distinct_gpus": len({(r[1], r[0]) for r in rows}), "distinct_hosts": len({r[1] for r in rows}),
            "rank_ms_min": round(lo, 3), "rank_ms_max": round(hi, 3), "rank_ms_mean": round(sum(ms) / len(ms), 3),
            "slowest_rank": ms.index(hi), "fastest_rank": ms.index(lo), "rank_ms_spread": round((hi - lo) / lo, 4) if lo > 0 else None,
            "host_cpus_min": min(r[3] for r in rows), "host_cpus_max": max(r[3] for r in rows),
            "numa_nodes_seen": len({r[4] for r in rows})}
