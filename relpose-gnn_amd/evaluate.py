"""Evaluation harness: the caller side of the hot path.

Restates what ``MultiModelTrainer.eval_RP`` does with the model outputs
(/root/reference/python/niantic/testing/test.py:180-286) so that the HIP module can be evaluated without the reference's
CLI: pick the reference edge of every graph, turn the predicted relative pose into the query's absolute pose, map the
log-quaternion back with ``qexp``, un-normalise the translation, accumulate translation / rotation errors, report
median and mean, and write the ``.npz`` the reference writes.  Differences by design: graphs are evaluated in
micro-batches (the reference uses batch_size=1, test.py:192; results per graph are independent of the batching), and the
stream can be sharded over ranks (``relpose_gnn_amd.shard``).  Host-side numpy in float64 like the reference.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterable, List, Optional, Sequence

import numpy as np
import torch

from . import _lib as _L
from .graph import Batch, Data


def qexp(v: np.ndarray) -> np.ndarray:
    """Exponential map R^3 -> unit quaternion (w, x, y, z): [cos|v|, sin|v|/|v| * v]   (pose_utils.py:340-348)."""
    v = np.asarray(v, dtype=np.float64)
    n = np.linalg.norm(v)
    return np.hstack((np.cos(n), np.sinc(n / np.pi) * v))


def quaternion_angular_error(q1: np.ndarray, q2: np.ndarray) -> float:
    """Angle in degrees between two unit quaternions: 2 acos(|<q1,q2>|)   (pose_utils.py:420-431)."""
    d = abs(float(np.dot(np.asarray(q2, dtype=np.float64), np.asarray(q1, dtype=np.float64))))
    d = min(1.0, max(-1.0, d))
    return 2.0 * np.arccos(d) * 180.0 / np.pi


def reference_edge(edges: np.ndarray, ref_node: int = 0) -> int:
    """Index of the ``ref_node``-th edge whose target is node 0, the query (test.py:227-229).  For the stored 8-node
    fully-connected graphs this is column 28, the edge 1 -> 0."""
    hits = np.argwhere(np.asarray(edges)[1] == 0)
    if hits.shape[0] <= ref_node:
        raise ValueError("graph has no edge into node 0: cannot derive the query pose")
    return int(hits[ref_node, 0])


def query_pose(rel_pose: np.ndarray, target: np.ndarray, edges: np.ndarray, pose_m, pose_s, ref_node: int = 0):
    """(pred[7], targ[7]) = (t, q) of the query node of ONE graph (test.py:227-251): absolute = ground truth of the
    reference edge's source node minus the predicted relative pose; rotation through qexp; translation * std + mean."""
    rel_pose, target = np.asarray(rel_pose, dtype=np.float64), np.asarray(target, dtype=np.float64)
    ref = reference_edge(edges, ref_node)
    out = target[np.asarray(edges)[0, ref]] - rel_pose[ref]
    pred = np.hstack((out[:3] * pose_s + pose_m, qexp(out[3:])))
    targ = np.hstack((target[0, :3] * pose_s + pose_m, qexp(target[0, 3:])))
    return pred, targ


@dataclass
class EvalResult:
    pred_poses: np.ndarray      # [G, 7]
    targ_poses: np.ndarray      # [G, 7]
    t_loss: np.ndarray          # [G] metres
    q_loss: np.ndarray          # [G] degrees

    @property
    def median_t(self) -> float:
        return float(np.median(self.t_loss))

    @property
    def median_q(self) -> float:
        return float(np.median(self.q_loss))

    def summary(self):
        """(median_t, mean_t, median_q, mean_q): the tuple eval_RP returns (test.py:286)."""
        return self.median_t, float(np.mean(self.t_loss)), self.median_q, float(np.mean(self.q_loss))


def errors(pred_poses: np.ndarray, targ_poses: np.ndarray) -> EvalResult:
    t_loss = np.asarray([np.linalg.norm(p - t) for p, t in zip(pred_poses[:, :3], targ_poses[:, :3])])
    q_loss = np.asarray([quaternion_angular_error(p, t) for p, t in zip(pred_poses[:, 3:], targ_poses[:, 3:])])
    return EvalResult(pred_poses, targ_poses, t_loss, q_loss)


def staging_workers(fp32_staging: bool, local_world: int = 1, available: Optional[int] = None, override: Optional[str] = None) -> int:
    """Staging threads of one rank's input pipeline.  One rank alone: 8 threads reach the ~40 GB/s a single fp32 stream needs, 16
    the bf16 stream's ~55 GB/s (memcpy for fp32 staging, rpg_host_f32_to_bf16 for bf16 staging).  But the host's copy rate PEAKS
    at ~16 threads IN ALL and collapses beyond (2 x EPYC 9575F, profiles/r5_stage_scale_*.jsonl: numpy copies 273 GB/s with 16
    threads, 81 with 64, 49 with 128; eight ranks' pipelines together stage 55 GB/s with 16 threads each, 73 with 8, 136 with 4,
    200 with TWO), so the ranks of one host share that budget: 16 // local_world threads per rank, at least 2 -- never more than
    the CPUs the process may run on.  RPG_STAGE_WORKERS (``override``) fixes the count."""
    import os
    own = 8 if fp32_staging else 16
    share = max(2, 16 // max(1, int(local_world)))
    if available is None:
        available = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 2) // 2
    forced = int((os.environ.get("RPG_STAGE_WORKERS", "0") if override is None else override) or 0)
    return max(1, min(forced or min(own, share), max(1, int(available))))


class _InputPipeline:
    """Host -> device staging of the node images of an evaluation stream, double buffered.

    The reference's loader hands over one graph at a time from pinned memory (``DataLoader(..., pin_memory=True)``,
    testing/test.py:193) and copies it with ``data.to(device)`` (:211).  Here a micro-batch of graphs is collated straight
    into one of TWO pinned host buffers (a ``copy_`` per graph: no ``torch.cat`` temporary, no pageable H2D), sent to one of
    two device buffers on a dedicated copy stream, and the forward of micro-batch i runs while micro-batch i+1 is being
    collated and copied.  Events order the three parties: ``sent[k]`` (copy stream: buffer k is on the device; the host may
    refill the pinned half, the compute stream may read the device half) and ``used[k]`` (compute stream: the forward that
    read device buffer k is done; the copy stream may overwrite it).  Graphs whose ``x`` is already pinned skip the staging
    copy; graphs already on the device skip the pipeline altogether (``evaluate_stream`` collates them on the device)."""

    def __init__(self, device, rows: int, row_floats: int, dtype=torch.float32, local_world: int = 1):
        # dtype = torch.bfloat16: the images are ROUNDED TO bf16 WHILE THEY ARE STAGED (the bf16 encoder rounds its fp32 input
        # first thing, so the forward is bit-identical) and the H2D copy is half the size -- at 256x341 the fp32 copy of a
        # 64-graph micro-batch (537 MB, ~13.7 ms at the 39 GB/s this host reaches) outlasts the bf16 forward (9 ms)
        self.device = device
        self.dtype = dtype
        self.copy_stream = torch.cuda.Stream(device=device)
        self.host = [torch.empty((rows, row_floats), dtype=dtype, pin_memory=True) for _ in range(2)]
        self.host_np = [t.numpy() for t in self.host] if dtype == torch.float32 else None
        import os
        from concurrent.futures import ThreadPoolExecutor
        self.workers = staging_workers(dtype == torch.float32, local_world)
        self.pool = ThreadPoolExecutor(max_workers=self.workers) if self.workers > 1 else None
        self.dev = [torch.empty((rows, row_floats), dtype=dtype, device=device) for _ in range(2)]
        self.sent = [torch.cuda.Event() for _ in range(2)]
        self.used = [torch.cuda.Event() for _ in range(2)]
        self.n_sent = [0, 0]
        self.staged_bytes = 0       # pageable -> pinned through the staging threads
        self.direct_bytes = 0       # sent straight out of the caller's own pinned tensors (no staging copy)

    def fits(self, rows: int, row_floats: int, dtype=torch.float32) -> bool:
        return rows <= self.host[0].shape[0] and row_floats == self.host[0].shape[1] and dtype == self.dtype

    def stage(self, k: int, chunk) -> torch.Tensor:
        """Collate the chunk's node images into pinned buffer k and enqueue the H2D copy; returns the device view."""
        return self.stage_end(self.stage_begin(k, chunk))

    def stage_begin(self, k: int, chunk):
        """First half of ``stage``: hand the chunk's pageable -> pinned copies (or bf16 roundings) to the staging threads and
        return at once; ``stage_end`` waits for them and enqueues the H2D copy.  Between the two the caller may do other host
        work (round 6: ``evaluate_stream`` post-processes the previous micro-batch there -- the copies release the GIL)."""
        if self.n_sent[k]:
            self.sent[k].synchronize()                 # the previous copy out of pinned buffer k has left the host
        host, dev, off = self.host[k], self.dev[k], 0
        direct = []                                    # graphs whose x is already pinned: copied from where they are
        jobs = []
        for g in chunk:
            n = g.x.shape[0]
            if g.x.is_pinned() and g.x.dtype == self.dtype:
                direct.append((off, n, g.x))
                self.direct_bytes += g.x.numel() * g.x.element_size()
            else:
                jobs.append((off, n, g.x))
                self.staged_bytes += g.x.numel() * host.element_size()
            off += n
        if jobs:
            # pageable -> pinned by a few worker threads (numpy releases the GIL; one memcpy stream moves ~5 GB/s on the GPU
            # host, measured r3: 649 graphs/s single-threaded against 1808 with resident images at 256x341)
            host_np = self.host_np[k] if self.host_np is not None else None
            cvt = _L.lib().rpg_host_f32_to_bf16 if host_np is None else None

            def copy_some(part):
                for o, n, src in part:
                    if host_np is not None:
                        np.copyto(host_np[o:o + n], src.detach().numpy())
                    else:
                        # fp32 -> bf16 (round to nearest even) on the way, in the library's host helper (ctypes releases the GIL)
                        t = src.detach()
                        if t.dtype != torch.float32 or not t.is_contiguous():
                            host[o:o + n].copy_(t)
                        else:
                            _L.check(cvt(t.data_ptr(), host[o:o + n].data_ptr(), t.numel()), "host_f32_to_bf16")
            w = min(len(jobs), self.workers)
            if w <= 1:
                copy_some(jobs)
                futs = []
            else:
                futs = [self.pool.submit(copy_some, jobs[i::w]) for i in range(w)]
        else:
            futs = []
        return k, off, direct, futs

    def stage_end(self, handle) -> torch.Tensor:
        k, off, direct, futs = handle
        host, dev = self.host[k], self.dev[k]
        for f in futs:
            f.result()
        with torch.cuda.stream(self.copy_stream):
            if self.n_sent[k]:
                self.copy_stream.wait_event(self.used[k])      # the forward that read device buffer k has finished
            if direct:
                lo = 0
                for o, n, src in direct:
                    if o > lo:
                        dev[lo:o].copy_(host[lo:o], non_blocking=True)
                    dev[o:o + n].copy_(src, non_blocking=True)
                    lo = o + n
                if off > lo:
                    dev[lo:off].copy_(host[lo:off], non_blocking=True)
            else:
                dev[:off].copy_(host[:off], non_blocking=True)
            self.sent[k].record(self.copy_stream)
        self.n_sent[k] += 1
        return dev[:off]

    def acquire(self, k: int) -> None:
        torch.cuda.current_stream().wait_event(self.sent[k])

    def release(self, k: int) -> None:
        self.used[k].record(torch.cuda.current_stream())


def _collate_on_device(chunk: Sequence[Data], x_dev: torch.Tensor, device) -> Batch:
    """The PyG collation of ``chunk`` (graph.Batch.from_data_list) with the node images already on the device in ``x_dev``:
    only the index tensors are built on the host (a few KB) and copied."""
    eis, bs, off = [], [], 0
    for gi, g in enumerate(chunk):
        n = g.x.shape[0]
        eis.append(g.edge_index + off)
        bs.append(torch.full((n,), gi, dtype=torch.int64))
        off += n
    out = Batch(x=x_dev, edge_index=torch.cat(eis, 1).to(device, non_blocking=True), y=None, edge_attr=None,
                batch=torch.cat(bs, 0).to(device, non_blocking=True))
    out.graph_sizes = ([g.x.shape[0] for g in chunk], [int(g.edge_index.shape[1]) for g in chunk])
    return out


class _MicroBatchRunner:
    """Launch side of the evaluation pipeline, shared by ``evaluate_stream`` and ``lookahead.Lookahead``: collate a chunk of
    single-graph ``Data`` objects (through the pinned double buffers of ``_InputPipeline`` when the images live on the host
    and the model on the GPU), run ONE forward over it and start the asynchronous copy of the poses (and of a model-built
    edge list) into pinned host memory.  ``launch`` returns without waiting; ``item.ev`` marks the poses' arrival."""

    def __init__(self, model, device, micro_batch: int, h2d_dtype=torch.float32, local_world: int = 1, want_abs: bool = False,
                 pinned_direct: bool = True):
        self.model, self.device, self.micro_batch = model, device, int(micro_batch)
        self.h2d_dtype, self.local_world, self.want_abs = h2d_dtype, local_world, want_abs
        # pinned_direct (when the caller did not force a staging dtype): a micro-batch whose fp32 images ALL sit in pinned host
        # memory -- what the reference's DataLoader(pin_memory=True) delivers, testing/test.py:193 -- goes to the device as it is,
        # straight from the loader's tensors.  For the fp32 model always (there is nothing to do on the host).  For a model that
        # takes host-rounded bf16 images it depends on what the rank can spend: the rounding pass halves the H2D bytes, and a Gen5
        # x16 link carries 51 GB/s = 6.1 k graphs/s of fp32 256 x 341 images where the bf16 forward does 8 k; with the staging
        # overlapped with the post-processing (round 6) 16 rounding threads deliver 7.8 k graphs/s, so a rank with >= 8 staging
        # threads (one or two ranks per host: staging_workers) rounds, and a rank with its 2-4 threads of an 8-rank host (2 threads
        # round ~27 ms per micro-batch) sends the loader's tensors as they are.
        self.pinned_direct = bool(pinned_direct) and not (h2d_dtype != torch.float32 and staging_workers(False, local_world) >= 8)
        self.on_gpu = torch.device(device).type == "cuda"
        self.pipes = {}                    # staging dtype -> _InputPipeline (at most two: the configured dtype, fp32 for pinned sources)
        self.n_batches = 0
        self.h2d_bytes = 0
        self._prefetched = None            # (chunk, pipe, stage_begin handle) of the chunk the next launch() is expected to get

    @property
    def pipe(self) -> Optional[_InputPipeline]:
        return self.pipes.get(self.h2d_dtype) or next(iter(self.pipes.values()), None)

    def pipe_stats(self) -> dict:
        ps = list(self.pipes.values())
        return {"staged_bytes": sum(p.staged_bytes for p in ps), "direct_bytes": sum(p.direct_bytes for p in ps),
                "staging_workers": max((p.workers for p in ps), default=0)}

    def _begin_staging(self, k: int, chunk):
        """Pick (or build) the chunk's pipeline and start its staging copies: -> (pipe, handle of _InputPipeline.stage_begin)."""
        device = self.device
        rows, width = sum(g.x.shape[0] for g in chunk), int(chunk[0].x.shape[1])
        dtype = self.h2d_dtype
        if self.pinned_direct and dtype != torch.float32 and all(g.x.dtype == torch.float32 and g.x.is_pinned() for g in chunk):
            dtype = torch.float32                               # the loader's own pinned fp32 tensors: no rounding pass, no staging copy
        pipe = self.pipes.get(dtype)
        if pipe is None or not pipe.fits(rows, width, dtype):
            if pipe is not None:
                torch.cuda.synchronize(device)                  # a larger buffer pair replaces one that is in flight
            cap = max(rows, max(g.x.shape[0] for g in chunk) * self.micro_batch)
            pipe = self.pipes[dtype] = _InputPipeline(torch.device(device), cap, width, dtype, self.local_world)
        return pipe, pipe.stage_begin(k, chunk)

    def prefetch(self, chunk: Sequence[Data]) -> None:
        """Start staging the chunk the NEXT ``launch`` call will get (its pageable -> pinned copies / bf16 roundings run on the
        staging threads while the caller does other host work); ``launch`` picks the result up if it is handed the same chunk."""
        if self._prefetched is not None or not chunk or not self.on_gpu or any(g.x.is_cuda for g in chunk):
            return
        pipe, handle = self._begin_staging(self.n_batches & 1, chunk)
        self._prefetched = (chunk, pipe, handle)

    def launch(self, chunk: Sequence[Data]):
        """-> (chunk, host_rel, host_ei | None, ev | None, host_abs | None, batch): batch = the collated micro-batch the forward
        read (batch.x: the chunk's node images, rows in chunk order; batch.edge_index: batch node ids)."""
        device, model = self.device, self.model
        k = self.n_batches & 1
        self.n_batches += 1
        staged = self.on_gpu and not any(g.x.is_cuda for g in chunk)
        pre, self._prefetched = self._prefetched, None
        if pre is not None and (not staged or len(pre[0]) != len(chunk) or any(a is not b for a, b in zip(pre[0], chunk))):
            pre[1].stage_end(pre[2])                                # a prefetch for another chunk: complete it (keeps the buffers' events in order), then drop it
            pre = None
        if staged:
            pipe, handle = (pre[1], pre[2]) if pre is not None else self._begin_staging(k, chunk)
            x_dev = pipe.stage_end(handle)
            self.h2d_bytes += x_dev.numel() * x_dev.element_size()
            batch = _collate_on_device(chunk, x_dev, device)
            pipe.acquire(k)
        else:
            if self.on_gpu and not all(g.x.is_cuda for g in chunk):
                # a chunk that mixes device- and host-resident graphs: the host ones go over one by one, the
                # collation then happens on the device (torch.cat would refuse mixed devices)
                batch = Batch.from_data_list([g.to(device, non_blocking=True) for g in chunk])
            else:
                batch = Batch.from_data_list(chunk).to(device, non_blocking=True)
        ab, rel, edge_index = model(batch)
        if staged:
            pipe.release(k)
        # a model-built edge list (kNN graph: the reference's default --knn 4, test.py:308, posenet.py:1047-1048) comes
        # back instead of the stored one: it travels to the host with the poses and is cut per graph by the consumer
        model_built = edge_index is not batch.edge_index
        host_abs = None
        if self.on_gpu:
            host = torch.empty(rel.shape, dtype=rel.dtype, pin_memory=True)
            host.copy_(rel, non_blocking=True)
            if self.want_abs:
                host_abs = torch.empty(ab.shape, dtype=ab.dtype, pin_memory=True)
                host_abs.copy_(ab, non_blocking=True)
            host_ei = None
            if model_built:
                host_ei = torch.empty(edge_index.shape, dtype=edge_index.dtype, pin_memory=True)
                host_ei.copy_(edge_index, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        else:
            host, host_ei, ev = rel, (edge_index if model_built else None), None
            host_abs = ab if self.want_abs else None
        return chunk, host, host_ei, ev, host_abs, batch


def edges_per_graph(ei: np.ndarray, sizes: Sequence[int]):
    """Cut a model-built edge list [2, E] of a collated micro-batch at graph boundaries: -> (first node of every graph,
    [column indices of graph k]); the batch order of the columns is kept (test.py:227 takes the FIRST edge into the node)."""
    first = np.concatenate([[0], np.cumsum(np.asarray(sizes))])
    gid = np.searchsorted(first, ei[1], side="right") - 1              # graph of every edge (by its target node)
    return first, [np.flatnonzero(gid == k) for k in range(len(sizes))]


@torch.no_grad()
def evaluate_stream(model, graphs: Sequence[Data], device, micro_batch: int = 64, pose_m=(0.0, 0.0, 0.0),
                    pose_s=(1.0, 1.0, 1.0), ref_node: int = 0, rank: int = 0, world: int = 1,
                    stats: Optional[dict] = None, bf16_input: Optional[bool] = None) -> EvalResult:
    """Run ``model`` over a stream of single-graph ``Data`` objects (x, edge_index, y) and post-process like test.py.
    With world > 1 every rank evaluates its contiguous block (shard_range) and the [G,7] rows are all-gathered.

    The loop is a three-stage software pipeline: while the GPU runs the forward of micro-batch i, the host collates
    micro-batch i+1 into pinned memory and the copy stream sends it (``_InputPipeline``; graphs that already live on the
    device are collated there instead), and once both are enqueued the host post-processes micro-batch i-1, whose relative
    poses have come back through an asynchronous copy -- so neither the H2D transfer of the images (537 MB per 64 graphs
    at 256x341), nor the D2H of the poses, nor the numpy work of test.py:213-251 leaves the GPU idle.
    ``stats`` (optional dict) receives ``h2d_bytes`` (bytes sent through the input pipeline), ``staged_bytes`` (of those: copied pageable -> pinned
    by the staging threads), ``direct_bytes`` (sent straight out of the caller's pinned tensors), ``staging_workers``, ``micro_batches`` and
    ``local_seconds`` (this rank's block, before the all-gather).  ``bf16_input`` (default: whatever the model
    accepts, i.e. True for the bf16 encoder with its fused stem): host-resident images are rounded to bf16 while they are staged --
    except, by default, micro-batches whose fp32 images all sit in PINNED memory (the reference's loader, test.py:193): those are sent
    as they are, with no host pass at all (``bf16_input=True`` forces the rounding pass for them too).  Same poses either way:
    the bf16 encoder rounds fp32 input first thing."""
    from .shard import rank_host_slice
    if world > 1 and torch.device(device).type == "cuda":
        import os
        # one process per GPU on a shared host: this rank's staging threads and pinned buffers stay on its share of the
        # cores / its GPU's NUMA node for the duration of the call (shard.rank_host_slice puts the caller's mask back on
        # return; a process its entry script has bound already is left as it is; RPG_BIND_RANKS=0 switches it off).  LOCAL_*
        # from the launcher when it set them (ranks of other nodes do not share this host)
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
        with rank_host_slice(int(os.environ.get("LOCAL_RANK", rank)), local_world, torch.device(device).index):
            return _evaluate_stream(model, graphs, device, micro_batch, pose_m, pose_s, ref_node, rank, world, stats, bf16_input, local_world)
    return _evaluate_stream(model, graphs, device, micro_batch, pose_m, pose_s, ref_node, rank, world, stats, bf16_input, 1)


def _evaluate_stream(model, graphs, device, micro_batch, pose_m, pose_s, ref_node, rank, world, stats, bf16_input, local_world):
    from .shard import gather_rows, shard_counts, shard_range
    pose_m, pose_s = np.asarray(pose_m, dtype=np.float64), np.asarray(pose_s, dtype=np.float64)
    lo, hi = shard_range(len(graphs), rank, world)
    preds: List[np.ndarray] = []
    targs: List[np.ndarray] = []
    # the bf16 encoder takes its node images in bf16 (rounded while they are staged: half the H2D bytes, identical results)
    h2d_dtype = torch.bfloat16 if (bf16_input if bf16_input is not None else getattr(model, "accepts_bf16_input", False)) else torch.float32
    runner = _MicroBatchRunner(model, device, micro_batch, h2d_dtype, local_world, pinned_direct=bf16_input is None)

    def finish(item):
        chunk, host, host_ei, ev = item[:4]
        if ev is not None:
            ev.synchronize()
        check = getattr(model, "check_edge_index", None)
        if check is not None:
            check(wait=False)  # this batch's bad-edge counters were copied before `ev`: look, do not wait for the NEXT batch
        rel = host.numpy()
        if host_ei is not None:
            ei = host_ei.numpy()
            first, per_graph = edges_per_graph(ei, [g.num_nodes for g in chunk])
            for k, g in enumerate(chunk):
                cols = per_graph[k]
                p, t = query_pose(rel[cols], g.y.cpu().numpy(), ei[:, cols] - first[k], pose_m, pose_s, ref_node)
                preds.append(p)
                targs.append(t)
            return
        e0 = 0
        for g in chunk:
            e = g.edge_index.shape[1]
            p, t = query_pose(rel[e0:e0 + e], g.y.cpu().numpy(), g.edge_index.cpu().numpy(), pose_m, pose_s, ref_node)
            preds.append(p)
            targs.append(t)
            e0 += e

    import time
    t_local = time.perf_counter()
    # Round 6: while the main thread post-processes micro-batch i - 1 (test.py:213-251 restated: ~3 ms of small numpy calls per
    # 64 graphs), the staging threads already copy / round micro-batch i + 1 into its pinned buffer (they release the GIL); its
    # H2D copy is then enqueued first thing in the next iteration, under the forward of micro-batch i.  Before, staging, launch and
    # post-processing took turns on the main thread and the bf16 stream at 256 x 341 was bound by that loop (9.5 ms per micro-batch
    # against 7.8 ms of GPU work).  Same launches, same order, same numbers.
    def chunk_at(b0):
        return [graphs[i] for i in range(b0, min(hi, b0 + micro_batch))]

    pending = None
    nxt = chunk_at(lo) if lo < hi else []
    for b0 in range(lo, hi, micro_batch):
        chunk = nxt
        item = runner.launch(chunk)
        nxt = chunk_at(b0 + micro_batch) if b0 + micro_batch < hi else []
        if nxt:
            runner.prefetch(nxt)
        if pending is not None:
            finish(pending)
        pending = item
    if pending is not None:
        finish(pending)
    if getattr(model, "check_edge_index", None) is not None:
        model.check_edge_index()                       # everything has been issued: wait for the last report
    if stats is not None:
        stats["local_seconds"] = time.perf_counter() - t_local      # this rank's own block, before the all-gather
        stats["h2d_bytes"] = runner.h2d_bytes
        stats["micro_batches"] = runner.n_batches
        # bytes that went pageable -> pinned through the staging threads / straight out of the caller's own pinned tensors
        stats.update(runner.pipe_stats())
    pred = np.stack(preds) if preds else np.zeros((0, 7))
    targ = np.stack(targs) if targs else np.zeros((0, 7))
    import torch.distributed as dist
    grouped = dist.is_available() and dist.is_initialized() and dist.get_world_size() == world
    if grouped and world == 1:
        # world size 1: only where the group's backend can take this device's tensors (an NCCL group cannot all-gather the CPU
        # tensors of a device="cpu" run -- ADVICE r5; before round 5 world = 1 never reached a collective)
        grouped = torch.device(device).type == "cuda" or dist.get_backend() != "nccl"
    if world > 1 or grouped:
        # under a process group the collective runs at ANY world size, 1 included: a one-GPU box then executes the same RCCL
        # all-gather the 4- / 8-GPU stream does (tools/eval_stream.py under torch.distributed.run)
        both = torch.from_numpy(np.concatenate([pred, targ], 1)).to(device)
        both = gather_rows(both, shard_counts(len(graphs), world), always=True).cpu().numpy()
        pred, targ = both[:, :7], both[:, 7:]
    return errors(pred, targ)


def seven_scenes_rel_paths(file_list: Sequence, dataset_filenames: Sequence, dataset_dir) -> List[str]:
    """The ``rel_path`` column eval_RP records for 7-Scenes (test.py:255-260): graph file ``data_<linear id>.pt`` ->
    ``dataset_filenames[linear id]`` (the colour frames of the TestSplit sequences in order, test.py:103-114) relative to
    the dataset directory."""
    import os
    out = []
    for fname in file_list:
        stem = os.path.splitext(os.path.basename(str(fname)))[0]
        linear_id = int(stem.split("_")[-1])
        out.append(os.path.relpath(str(dataset_filenames[linear_id]), str(dataset_dir)))
    return out


def save_poses(path, result: EvalResult, rel_paths: Iterable) -> None:
    """The .npz of test.py:38-42 (same field names, same length assertion).  ``rel_paths``: one entry per graph, for
    7-Scenes ``seven_scenes_rel_paths(...)``."""
    rel_paths = [str(p) for p in rel_paths]
    assert len(rel_paths) == len(result.pred_poses), \
        f"len(rel_paths): {len(rel_paths)} != {len(result.pred_poses)} len(pred_poses)"
    np.savez(path, rel_path=rel_paths, abs_t=result.pred_poses[:, :3], abs_q=result.pred_poses[:, 3:],
             targ_t=result.targ_poses[:, :3], targ_q=result.targ_poses[:, 3:])


def load_pose_stats(path):
    """Cambridge translation mean / std file, two rows of three numbers (test.py:126-130)."""
    m, s = np.loadtxt(path)
    return m, s
