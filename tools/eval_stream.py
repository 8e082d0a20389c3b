#!/usr/bin/env python3
"""BASELINE.json configs[3]/[4]: an evaluation-shaped STREAM of query graphs through the HIP module, sharded over the
ranks of one node and collated with one all-gather (RCCL) of the derived query poses.

    python tools/eval_stream.py --graphs 2000 --shape 256x341                      # 7-Scenes 'chess' test split size
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29500 \\
        tools/eval_stream.py --graphs 2000 --shape 256x341
    ... --graphs 17000 --encoder-dtype bf16                                        # configs[4]: all seven scenes, bf16 encoder

Pixels are synthetic (no dataset here); the edge lists are the reference's fully-connected 8-node graphs; every graph
goes through the same post-processing as testing/test.py:213-251.  The loop is the PRODUCT's: this script only builds the
stream (host-resident single-graph `Data` objects, as the reference's loader delivers them, test.py:193,211) and calls
`relpose_gnn_amd.evaluate.evaluate_stream`, which micro-batches (the reference evaluates with batch_size=1; graphs are
independent), stages the node images through pinned double buffers on a copy stream, and overlaps D2H + post-processing.
`--input resident` keeps the images on the device (no H2D) for comparison.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL (bench.py explains); before the HIP runtime starts

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=2000)
    ap.add_argument("--shape", default="256x341")
    ap.add_argument("--micro-batch", type=int, default=64, help="graphs per forward (64: +5 % over 32 at 256x341 with the persistent Winograd kernel)")
    ap.add_argument("--encoder-dtype", choices=("f32", "bf16"), default="f32")
    ap.add_argument("--gnn-dtype", choices=("f32", "bf16"), default="f32")
    ap.add_argument("--knn", type=int, default=-1, help="the reference's --knn (test.py:308 defaults to 4): the model rebuilds "
                    "the graph from the encoder features (posenet.py:1047-1048); -1 = the stored fully-connected edges")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE", help="rpg_set_tuning(KEY, VALUE) before the run (A/B)")
    ap.add_argument("--input", choices=("host", "pinned", "resident"), default="host",
                    help="where the graphs' node images live when the stream starts: pageable host memory (a plain loader), pinned "
                         "host memory (DataLoader(pin_memory=True), test.py:193), or the device (no H2D: the comparison point)")
    ap.add_argument("--h2d", choices=("auto", "f32", "bf16"), default="auto",
                    help="staging dtype of host-resident images: auto = bf16 where the model takes it (bf16 encoder: rounded while staged, half the H2D bytes, identical results)")
    ap.add_argument("--pool", type=int, default=96, help="distinct graphs' worth of synthetic pixels that the stream cycles through")
    args = ap.parse_args()
    h, w = (int(v) for v in args.shape.split("x"))
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    # under a launcher (torchrun environment) the process group is RCCL at ANY world size, 1 included: a one-GPU box then runs
    # the same all-gather / barrier / all_reduce the 4- and 8-GPU streams run (tests/test_hip_00_rccl.py)
    under_launcher = "WORLD_SIZE" in os.environ
    if under_launcher:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        if world > 1:
            # process-long host placement is the ENTRY SCRIPT's decision (it knows LOCAL_RANK); evaluate_stream itself only
            # binds for the duration of a call and leaves a process bound here alone
            from relpose_gnn_amd.shard import bind_rank_to_host_slice
            bind_rank_to_host_slice(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)), local)
    import relpose_gnn_amd.synth as S
    from relpose_gnn_amd import evaluate as E
    from relpose_gnn_amd.posenet import PoseNetX_R2
    from relpose_gnn_amd.resnet import resnet34

    D = 2048
    model = PoseNetX_R2(resnet34(), droprate=0.0, pretrained=False, feat_dim=D, edge_feat_dim=D, node_dim=D,
                        input_img_height=h, use_gnn=True, knn=args.knn, use_AP=True, gnn_recursion=2)
    model.load_state_dict(S.synth_state_dict(S.posenet_r2_param_shapes(D, D, D), seed=1))
    model = model.to(dev).eval()
    model.encoder_dtype = args.encoder_dtype
    model.gnn_dtype = args.gnn_dtype
    from relpose_gnn_amd import ops
    for kv in args.tune:
        k, v = kv.split("=")
        ops.set_tuning(int(k), int(v))

    # The stream: `--graphs` single-graph Data objects as a loader delivers them -- x on the HOST (pageable by default,
    # `--input pinned` = a pin_memory loader, `--input resident` = already on the device: the no-H2D comparison point).
    # Pixels are synthetic; `--pool` distinct graphs' worth of them are cycled (2000 distinct 256x341 graphs would be 17 GB).
    from relpose_gnn_amd.graph import Data, fc_edge_index
    gen = torch.Generator().manual_seed(77 + rank)
    ei8 = fc_edge_index(8)
    pool = []
    for i in range(min(args.pool, args.graphs)):
        xi = torch.randn((8, 3 * h * w), generator=gen)
        if args.input == "pinned":
            xi = xi.pin_memory()
        elif args.input == "resident":
            xi = xi.to(dev)
        pool.append((xi, torch.randn((8, 6), generator=gen) * 0.3))            # pixels, ground-truth poses (t, log q)
    graphs = [Data(x=pool[i % len(pool)][0], edge_index=ei8, y=pool[i % len(pool)][1]) for i in range(args.graphs)]
    mb = args.micro_batch

    bfin = None if args.h2d == "auto" else args.h2d == "bf16"
    E.evaluate_stream(model, graphs[: min(2 * mb, len(graphs))], dev, micro_batch=mb, bf16_input=bfin)      # warm-up (packing, workspaces, staging buffers)
    if under_launcher:
        dist.barrier()
    torch.cuda.synchronize()
    stats = {}
    t0 = time.perf_counter()
    res = E.evaluate_stream(model, graphs, dev, micro_batch=mb, rank=rank, world=world, stats=stats, bf16_input=bfin)   # THE product loop
    torch.cuda.synchronize()
    if under_launcher:
        dist.barrier()
    dt = time.perf_counter() - t0
    from relpose_gnn_amd.shard import rank_report
    report = rank_report(dev, stats.get("local_seconds", dt), 1)          # collective; this rank's own clock over its block of the stream
    if under_launcher:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        assert len(res.pred_poses) == args.graphs and np.isfinite(res.pred_poses).all()
        print(json.dumps({"workload": f"eval-shape stream through relpose_gnn_amd.evaluate.evaluate_stream: {args.graphs} 8-node FC graphs, "
                                      f"{h}x{w}, encoder {args.encoder_dtype}, GNN Linears {args.gnn_dtype}, knn {args.knn}, micro-batch {mb}, "
                                      f"node images {args.input} ({'pinned double-buffered H2D on a copy stream' if args.input != 'resident' else 'no H2D'}"
                                      f"{', staged as bf16' if args.input != 'resident' and (bfin if bfin is not None else model.accepts_bf16_input) else ''}), "
                                      "D2H + test.py post-processing per graph included",
                          "input": args.input, "n_gpus": world,
                          # what the collectives observed (shard.rank_report): ranks that took part, distinct GPUs / hosts / NUMA nodes,
                          # each rank's own milliseconds over ITS block of the stream (rank_ms_min / _max / _mean, slowest_rank), CPUs per rank
                          **report,
                          "staged_gb": round(stats.get("staged_bytes", 0) / 1e9, 3), "direct_gb": round(stats.get("direct_bytes", 0) / 1e9, 3),
                          "staging_workers": stats.get("staging_workers"), "graphs": args.graphs, "seconds": round(dt, 3),
                          "graphs_per_s": round(args.graphs / dt, 1),
                          "h2d_gb_per_s": round(stats.get("h2d_bytes", 0) / dt / 1e9, 2)}), flush=True)
    if under_launcher:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
