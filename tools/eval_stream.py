#!/usr/bin/env python3
"""BASELINE.json configs[3]/[4]: an evaluation-shaped STREAM of query graphs through the HIP module, sharded over the
ranks of one node and collated with one all-gather (RCCL) of the derived query poses.

    python tools/eval_stream.py --graphs 2000 --shape 256x341                      # 7-Scenes 'chess' test split size
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29500 \\
        tools/eval_stream.py --graphs 2000 --shape 256x341
    ... --graphs 17000 --encoder-dtype bf16                                        # configs[4]: all seven scenes, bf16 encoder

Pixels are synthetic (no dataset here); the edge lists are the reference's fully-connected 8-node graphs; every graph
goes through the same post-processing as testing/test.py:213-251 (`relpose_gnn_amd.evaluate`).  The reference evaluates
with batch_size=1; graphs are independent, so they are micro-batched here (--micro-batch).  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

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
    args = ap.parse_args()
    h, w = (int(v) for v in args.shape.split("x"))
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    import relpose_gnn_amd.synth as S
    from relpose_gnn_amd import evaluate as E
    from relpose_gnn_amd.graph import fc_batch
    from relpose_gnn_amd.posenet import PoseNetX_R2
    from relpose_gnn_amd.resnet import resnet34
    from relpose_gnn_amd.shard import gather_rows, shard_counts, shard_range

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

    lo, hi = shard_range(args.graphs, rank, world)
    mb = args.micro_batch
    gen = torch.Generator(device=dev).manual_seed(77 + rank)
    x = torch.randn((8 * mb, 3 * h * w), generator=gen, device=dev)           # pixel buffer reused by every micro-batch
    y = torch.randn((8 * mb, 6), generator=gen, device=dev) * 0.3              # ground-truth poses (t, log q)
    pm, ps = np.zeros(3), np.ones(3)

    ei_local = None

    def run(count):
        """Software-pipelined like evaluate.evaluate_stream: enqueue micro-batch i+1, then post-process micro-batch i."""
        nonlocal ei_local
        preds, pending, done = [], None, 0
        y_host = y.cpu().numpy()

        def finish(item):
            g, host, host_ei, ev = item
            ev.synchronize()
            model.check_edge_index(wait=False)           # this micro-batch's device-side index check landed with `ev`
            rel_c = host.numpy()
            if host_ei is not None:                      # model-built (kNN) edges: cut per graph by the target node's graph
                ei = host_ei.numpy()
                gid = ei[1] // 8
                for k in range(g):
                    cols = np.flatnonzero(gid == k)
                    p, _ = E.query_pose(rel_c[cols], y_host[8 * k: 8 * (k + 1)], ei[:, cols] - 8 * k, pm, ps)
                    preds.append(p)
                return
            for k in range(g):
                p, _ = E.query_pose(rel_c[56 * k: 56 * (k + 1)], y_host[8 * k: 8 * (k + 1)], ei_local, pm, ps)
                preds.append(p)

        while done < count:
            g = min(mb, count - done)
            batch = fc_batch(x[: 8 * g], 8, y[: 8 * g])
            if ei_local is None:
                ei_local = batch.edge_index[:, :56].cpu().numpy()
            _, rel, ei_out = model(batch)
            host = torch.empty(rel.shape, dtype=rel.dtype, pin_memory=True)
            host.copy_(rel, non_blocking=True)                                  # D2H as test.py:214 does, asynchronously
            host_ei = None
            if ei_out is not batch.edge_index:
                host_ei = torch.empty(ei_out.shape, dtype=ei_out.dtype, pin_memory=True)
                host_ei.copy_(ei_out, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            if pending is not None:
                finish(pending)
            pending = (g, host, host_ei, ev)
            done += g
        if pending is not None:
            finish(pending)
        model.check_edge_index()
        return np.stack(preds) if preds else np.zeros((0, 7))

    run(min(mb, hi - lo))                                                      # warm-up (packing, workspaces)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    local_pred = run(hi - lo)
    poses = gather_rows(torch.from_numpy(local_pred).to(dev), shard_counts(args.graphs, world)) if world > 1 else local_pred
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        assert len(poses) == args.graphs
        print(json.dumps({"workload": f"eval-shape stream: {args.graphs} 8-node FC graphs, {h}x{w}, encoder {args.encoder_dtype}, GNN Linears {args.gnn_dtype}, "
                                      f"knn {args.knn}, micro-batch {mb}, pipelined D2H + test.py post-processing per graph included",
                          "n_gpus": world, "graphs": args.graphs, "seconds": round(dt, 3),
                          "graphs_per_s": round(args.graphs / dt, 1)}), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
