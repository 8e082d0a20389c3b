#!/usr/bin/env python3
"""Host staging at scale (round 5): can eight ranks' input pipelines share one two-socket host?

Starts W processes (default 8) that each run ONLY the producer side of relpose_gnn_amd.evaluate._InputPipeline -- collate
micro-batches of pageable host images into pinned memory with the staging threads (memcpy for fp32, rpg_host_f32_to_bf16 for
bf16) and send them host -> device on a copy stream; no forward -- and reports the aggregate GB/s of pageable bytes consumed,

  * with every rank bound to its share of the host (shard.bind_rank_to_host_slice: the default of evaluate_stream(rank, world)),
  * and unbound (RPG_BIND_RANKS=0: threads and pinned buffers float over both sockets),
  * with and without the H2D copy (`--no-h2d`: the one PCIe link of a one-GPU box is shared by all W processes, so the
    staging-only rate is the host-side number; on an 8-GPU node every rank has its own link).

    python tools/stage_scale.py --ranks 8 --dtype bf16 --seconds 6        # prints one JSON line per configuration

Every process uses cuda:0 here (the box has one GPU).  The parent never touches the GPU; children are plain subprocesses."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(args):
    sys.path.insert(0, ROOT)
    import torch
    from relpose_gnn_amd import shard
    from relpose_gnn_amd.evaluate import _InputPipeline
    from relpose_gnn_amd.graph import Data, fc_edge_index
    rank, world = args.rank, args.ranks
    cpus = shard.bind_rank_to_host_slice(rank, world, 0) if args.bind else None
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    h, w = (int(v) for v in args.shape.split("x"))
    mb, nodes = args.micro_batch, 8
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    gen = torch.Generator().manual_seed(rank)
    ei = fc_edge_index(nodes)
    pool = [Data(x=torch.randn((nodes, 3 * h * w), generator=gen), edge_index=ei, y=None) for _ in range(args.pool)]   # pageable, first touched HERE (after binding)
    pipe = _InputPipeline(dev, mb * nodes, 3 * h * w, dtype, world)              # the product default: 16 // ranks staging threads (RPG_STAGE_WORKERS overrides)
    if args.no_h2d:
        class _NoCopy:                                   # staging only: the pinned buffer is filled, nothing is sent
            def copy_(self, *a, **k):
                return self
        pipe.dev = [type("D", (), {"__getitem__": lambda s, i: _NoCopy()})() for _ in range(2)]
    chunk = [pool[i % len(pool)] for i in range(mb)]
    for k in (0, 1):
        pipe.stage(k, chunk)
    torch.cuda.synchronize()
    # line up the ranks on a wall-clock deadline (no process group needed)
    while time.time() < args.start_at:
        time.sleep(0.001)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < args.seconds:
        k = n & 1
        pipe.stage(k, chunk)
        if not args.no_h2d:
            pipe.acquire(k)
            pipe.release(k)
        n += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    src_bytes = n * mb * nodes * 3 * h * w * 4
    print(json.dumps({"rank": rank, "batches": n, "seconds": dt, "pageable_gb_per_s": src_bytes / dt / 1e9,
                      "workers": pipe.workers, "cpus": len(cpus) if cpus else len(os.sched_getaffinity(0))}), flush=True)


def memcpy_probe(seconds: float):
    """Host characterisation: aggregate pageable -> pageable copy rate of T numpy threads (64-MB buffers, each thread its own pair;
    the GIL is released inside np.copyto), T = 1 .. 128: what the staging threads of ALL ranks can share on this host."""
    import threading
    import numpy as np
    out = []
    for threads in (1, 4, 16, 32, 64, 128):
        src = [np.ones(16 << 20, dtype=np.float32) for _ in range(threads)]
        dst = [np.empty_like(a) for a in src]
        counts = [0] * threads
        stop = time.perf_counter() + seconds

        def work(i):
            while time.perf_counter() < stop:
                np.copyto(dst[i], src[i])
                counts[i] += 1
        ts = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        dt = time.perf_counter() - t0
        rec = {"memcpy_threads": threads, "read_gb_per_s": round(sum(counts) * (64 << 20) / dt / 1e9, 1), "buffers_mb": 64}
        print(json.dumps(rec), flush=True)
        out.append(rec)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="bf16")
    ap.add_argument("--shape", default="256x341")
    ap.add_argument("--micro-batch", type=int, default=64)
    ap.add_argument("--pool", type=int, default=16)
    ap.add_argument("--seconds", type=float, default=6.0)
    ap.add_argument("--rank", type=int, default=-1)
    ap.add_argument("--bind", type=int, default=1)
    ap.add_argument("--no-h2d", action="store_true")
    ap.add_argument("--start-at", type=float, default=0.0)
    ap.add_argument("--workers", default="", help="comma-separated staging-thread counts per rank to sweep (RPG_STAGE_WORKERS); default: the pipeline's own")
    ap.add_argument("--only-ranks", action="store_true", help="skip the single-rank rows")
    ap.add_argument("--memcpy-probe", action="store_true", help="only the host copy-rate characterisation (no GPU, no ranks)")
    args = ap.parse_args()
    if args.memcpy_probe:
        return memcpy_probe(min(args.seconds, 3.0))
    if args.rank >= 0:
        return child(args)
    out = []
    sweep = [int(v) for v in args.workers.split(",") if v.strip()] or [0]
    for ranks in ([args.ranks] if args.only_ranks else sorted({1, args.ranks})):
      for workers in sweep:
        for no_h2d in ((True,) if args.workers else (True, False)):
            for bind in ((1,) if args.workers else (1, 0)):
                if ranks == 1 and bind == 0:
                    continue
                start = time.time() + 60.0 + 2.0 * ranks                   # children import torch (~seconds each) before the deadline
                procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--rank", str(r), "--ranks", str(ranks), "--dtype", args.dtype,
                                           "--shape", args.shape, "--micro-batch", str(args.micro_batch), "--pool", str(args.pool),
                                           "--seconds", str(args.seconds), "--bind", str(bind), "--start-at", str(start)] + (["--no-h2d"] if no_h2d else []),
                                          stdout=subprocess.PIPE, text=True,
                                          env=dict(os.environ, RPG_BIND_RANKS=str(bind), **({"RPG_STAGE_WORKERS": str(workers)} if workers else {}))) for r in range(ranks)]
                recs = []
                for p in procs:
                    o, _ = p.communicate(timeout=600)
                    recs += [json.loads(ln) for ln in o.splitlines() if ln.startswith("{")]
                agg = sum(r["pageable_gb_per_s"] for r in recs)
                rec = {"ranks": ranks, "dtype": args.dtype, "shape": args.shape, "bound": bool(bind), "h2d": not no_h2d,
                       "aggregate_pageable_gb_per_s": round(agg, 2), "per_rank_gb_per_s": [round(r["pageable_gb_per_s"], 2) for r in recs],
                       "graphs_per_s_equivalent": round(agg * 1e9 / (8 * 3 * int(args.shape.split('x')[0]) * int(args.shape.split('x')[1]) * 4), 1),
                       "workers_per_rank": recs[0]["workers"] if recs else None, "cpus_per_rank": recs[0]["cpus"] if recs else None,
                       "ranks_reporting": len(recs)}
                print(json.dumps(rec), flush=True)
                out.append(rec)
    return out


if __name__ == "__main__":
    main()
