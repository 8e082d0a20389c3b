#!/usr/bin/env python3
"""Benchmark of the relpose-gnn inference hot path on MI355X: graphs/sec on synthetic 8-node fully-connected
224x224 graphs (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

A step = one forward of PoseNetX_R2 (ResNet34 encoder -> FC-graph GNN -> pose heads, R3 dims: D=2048,
gnn_recursion=2, droprate=0, eval) over one batch of 32 graphs (256 images) per GPU -- BASELINE.json configs[1] --
with the inputs already resident in HBM; for N>1 each rank runs its own batch (weak scaling, graphs are independent)
and the predicted relative poses are all-gathered over RCCL every step.  Rank 0 prints ONE JSON line.

`python bench.py --gpus N` with N > 1 and no torchrun environment starts the N ranks itself (a child
`python -m torch.distributed.run`, spawned before this process touches the GPU) and relays their JSON line.

The line also carries
  roofline     for the dominant kernel (the Winograd F(4,3) f32-MFMA convolution): the FLOP its launches EXECUTE on the
               matrix pipe / their summed HIP-event durations, against the 157.3 TFLOP/s f32 matrix peak (frac <= 1); the
               algorithmic (direct-convolution) rate and the algorithmic / executed ratio are separate fields;
  cpu_baseline the CPU oracle (validated against the reference in this repo's build container) timed on this box's
               host cores on a bounded sample of the same workload (N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# RCCL shares device buffers between the ranks' processes through HIP IPC handles and the pool's host driver only supports
# the dmabuf flavour: without this the first collective of a multi-rank run fails with "hipIpcGetMemHandle: invalid
# argument".  Already exported on the GPU boxes; set here too (before the HIP runtime starts) for any other launcher.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

BF16_MATRIX_PEAK_TFLOPS = 2500.0    # dense bf16 MFMA (MI355X_MICROARCH.md)
F32_MATRIX_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
HBM_PEAK_GBS = 8000.0
NODES, IMG = 8, 224


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--graphs", type=int, default=32, help="graphs per step per GPU (configs[1]: 32)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the secondary BASELINE.json configs (configs[2] bf16 at 64 graphs, configs[3]/[4] evaluation streams at "
                         "256x341) that the default N=1 run measures after the headline and reports under `other_configs`")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=120.0,
                    help="budget of the whole CPU baseline leg (three thread counts x B = 1/8/32); 0 disables it")
    ap.add_argument("--no-kernel-timing", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-graph latency leg (profiling runs: keeps its launches out of the trace)")
    ap.add_argument("--launcher", action="store_true",
                    help="start the rank(s) through this script's own torch.distributed.run command line even for --gpus 1: the "
                         "launcher, RCCL initialisation and the per-step all-gather then run on a one-GPU box too")
    ap.add_argument("--streams", type=int, default=2, help="HIP streams per GPU the batch is spread over in the timed region")
    ap.add_argument("--gnn-dtype", choices=("f32", "bf16"), default="f32",
                    help="bf16: the GNN's Linears on the bf16 matrix pipe too (only meaningful with --encoder-dtype bf16)")
    ap.add_argument("--bf16-bk", type=int, default=0, help="K step of the bf16 conv kernel (32|64; 0 = library default)")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE",
                    help="rpg_set_tuning(KEY, VALUE) before the run, for A/B experiments (e.g. --tune 8=0: no Winograd split-K tail)")
    ap.add_argument("--schedule", default="", help="explicit stream schedule g0:g1:slot,... (experiments; default: --streams equal groups)")
    ap.add_argument("--encoder-dtype", choices=("f32", "bf16"), default="f32",
                    help="f32 = the headline configuration (configs[1]); bf16 = configs[2] (bf16 activations + MFMA conv)")
    return ap.parse_args()


CPU_THREADS_BEST = 32      # measured optimum of torch-CPU for this model on the 2 x EPYC 9575F host (tests/probes/cpu_threads_probe.py)


def host_sockets() -> int:
    """Sockets of the host per lscpu (0 = unknown)."""
    try:
        import subprocess
        out = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        f = {ln.split(":", 1)[0].strip(): ln.split(":", 1)[1].strip() for ln in out.splitlines() if ":" in ln}
        return int(f["Socket(s)"])
    except Exception:
        return 0


def physical_cores() -> int:
    """Physical cores of the host ("state the count from lscpu"); falls back to logical / 2."""
    try:
        import subprocess
        out = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        f = {ln.split(":", 1)[0].strip(): ln.split(":", 1)[1].strip() for ln in out.splitlines() if ":" in ln}
        n = int(f["Core(s) per socket"]) * int(f["Socket(s)"])
        if n > 0:
            return n
    except Exception:
        pass
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def cpu_worker(budget_s: float, threads: int):
    """One CPU-baseline worker = one thread count: the oracle forward (PyTorch CPU fp32) at B = 1, 8 and 32 graphs per
    forward, 1 warm-up + the median of 3 timed forwards each; a cell whose projected cost does not fit
    what is left of ``budget_s`` is recorded as skipped instead of run.  Prints one JSON line."""
    from oracle import posenet_ref as O              # the checker, timed here as the reported CPU baseline
    import relpose_gnn_amd.synth as S
    torch.set_num_threads(threads)
    sd = S.synth_state_dict(S.posenet_r2_param_shapes(), seed=1)
    t_start = time.perf_counter()
    cells, per_graph = [], None
    for g in (1, 8, 32):
        left = budget_s - (time.perf_counter() - t_start)
        if per_graph is not None and 4 * g * per_graph > left:
            cells.append({"threads": threads, "graphs_per_forward": g, "skipped": f"projected {4 * g * per_graph:.0f} s > {left:.0f} s left of this thread count's budget"})
            continue
        x = S.synth_images(NODES * g, IMG, IMG, seed=77)
        ei = O.batch_edge_index(NODES, g)
        O.posenet_forward(sd, x, ei, IMG, 2)             # warm-up
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            O.posenet_forward(sd, x, ei, IMG, 2)
            ts.append(time.perf_counter() - t0)
        med = sorted(ts)[1]
        per_graph = med / g
        cells.append({"threads": threads, "graphs_per_forward": g, "median_s": round(med, 4), "timed_forwards": 3,
                      "graphs_per_s": round(g / med, 3)})
    print(json.dumps({"cells": cells, "seconds": time.perf_counter() - t_start}), flush=True)


def cpu_baseline(budget_s: float):
    """SURVEY 8(d) protocol: the CPU oracle (validated against the reference in the build container: "port") with
    torch.set_num_threads(k) for k = all physical cores of this box, k = 8 (comparable with the build container) and
    k = 32 (the optimum measured on the 2 x EPYC 9575F GPU host: oneDNN convolutions are memory-bound and do not scale
    across the sockets -- 128 threads 3.3 graphs/s, 64: 7.1, 32: 12.1, 16: 10.7); B = 1 / 8 / 32 graphs per forward,
    1 warm-up + median of 3.  One child process per thread count, all BEFORE this process touches the GPU.  `value` = the
    fastest cell, `cores` = its thread count, the rest is under `sweep`.  ``budget_s`` bounds the whole leg."""
    import subprocess
    phys = physical_cores()
    counts = []
    for k in (min(CPU_THREADS_BEST, phys), min(8, phys), phys):      # cheapest first; the all-cores run gets what is left
        if k not in counts:
            counts.append(k)
    t0 = time.perf_counter()
    sweep, secs = [], 0.0
    for i, k in enumerate(counts):
        left = budget_s - (time.perf_counter() - t0)
        share = max(5.0, left / (len(counts) - i))
        env = dict(os.environ, OMP_NUM_THREADS=str(k), MKL_NUM_THREADS=str(k))
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(share), str(k)],
                             stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, text=True)
        try:
            out, _ = p.communicate(timeout=share * 3 + 240)
            r = json.loads(out.strip().splitlines()[-1])
            sweep += r["cells"]
            secs += r["seconds"]
        except Exception as exc:                                     # a baseline leg must not take the bench down
            p.kill()
            try:
                p.communicate(timeout=10)                            # reap the child (no zombie, pipes closed)
            except Exception:
                pass
            sweep.append({"threads": k, "skipped": f"worker failed: {type(exc).__name__}"})
    done = [c for c in sweep if "graphs_per_s" in c]
    if not done:
        return None
    best = max(done, key=lambda c: c["graphs_per_s"])
    return {"value": best["graphs_per_s"], "unit": "graphs/s", "cores": best["threads"], "kind": "port",
            "physical_cores": phys, "sockets": host_sockets(),
            "sample": f"oracle forward (torch {torch.__version__} CPU fp32) of {best['graphs_per_forward']} 8-node graphs "
                      f"(= {8 * best['graphs_per_forward']} images 224x224) per call, {best['threads']} threads, 1 warm-up + median "
                      f"of 3; fastest of the sweep threads in {counts} x graphs-per-forward in (1, 8, 32) on a {host_sockets()}-socket host with "
                      f"{phys} physical cores (no NUMA pinning: the all-cores rows lose to 8 / 32 threads because oneDNN's "
                      f"convolutions are memory-bound across the sockets); {secs:.0f} s of CPU work",
            "sweep": sweep}


def bf16_roofline(kt):
    """`roofline` object of a bf16-encoder run from the library's per-launch HIP-event timings (one-stream pass): the
    ALGORITHMIC FLOP of all encoder convolution launches / their summed durations against the dense bf16 matrix peak."""
    c = kt["conv"]
    tf = c["work"] / (c["ms"] * 1e-3) / 1e12
    return {"bound": "mfma", "achieved": round(tf, 2), "peak": BF16_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(tf / BF16_MATRIX_PEAK_TFLOPS, 4), "traffic": None,
            "kernel": "conv3x3_bf16_patch_kernel (every 3x3/s1 convolution: input patch resident in LDS, branch-free epilogue, "
                      "division-free tile set-up) / conv_bf16_dma_kernel (3x3/s2, 1x1/s2: implicit GEMM, operands by LDS-DMA, counted "
                      "vmcnt) / stem_pool_bf16_kernel (fused stem), all on v_mfma_f32_32x32x16_bf16",
            "launches": c["launches"], "avg_launch_ms": round(c["ms"] / c["launches"], 4),
            "what": "achieved = ALGORITHMIC FLOP of all the encoder's convolution launches (2 * pixels * Cout * "
                    "kh * kw * Cin; the stem as 7x7x3) / their summed HIP-event durations in the one-stream pass"}


def other_configs(model, args, dev):
    """The secondary BASELINE.json configurations, measured by the SAME process right after the headline (N = 1 only) so that
    the driver's JSON line carries them.  `value` of the line stays the fp32 configs[1] number.

      configs2_bf16_encoder / configs2_bf16_all   64 graphs x 8 x 224x224 per step, bf16 encoder (+ bf16 GNN Linears), inputs
                                                   resident, `--streams` streams; + the one-stream HIP-event pass -> `roofline`
      configs3_eval_stream_1gpu_host_fp32          2000 graphs of 8 x 256x341 in PAGEABLE host memory through the product loop
      configs4_eval_stream_1gpu_host_bf16          relpose_gnn_amd.evaluate.evaluate_stream (micro-batches of 64, pinned double-
                                                   buffered H2D on a copy stream, D2H + test.py post-processing): fp32 model /
                                                   bf16 encoder + bf16 GNN Linears with the images rounded to bf16 while staged
    The 4- / 8-GPU sharding of configs[3] / [4] is the driver's to launch (tools/eval_stream.py under torch.distributed.run)."""
    from relpose_gnn_amd import evaluate as E
    from relpose_gnn_amd import ops
    from relpose_gnn_amd.graph import Data, fc_batch, fc_edge_index
    out = {}
    steps = max(args.steps, 5)
    g2 = 64
    x2 = torch.randn((NODES * g2, 3 * IMG * IMG), generator=torch.Generator(device=dev).manual_seed(4321), device=dev)
    d2 = fc_batch(x2, NODES)
    try:
        for name, gnn in (("configs2_bf16_encoder", "f32"), ("configs2_bf16_all", "bf16")):
            model.encoder_dtype, model.gnn_dtype, model.hip_streams = "bf16", gnn, args.streams
            for _ in range(3):
                _, rel, _ = model(d2)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                _, rel, _ = model(d2)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            assert bool(torch.isfinite(rel).all())
            model.hip_streams = 1
            model(d2)
            ops.timing_read()
            ops.timing_enable(True)
            for _ in range(steps):
                model(d2)
            torch.cuda.synchronize()
            ops.timing_enable(False)
            kt = ops.timing_read()
            out[name] = {"value": round(g2 * steps / dt, 1), "unit": "graphs/s", "ms_per_step": round(1e3 * dt / steps, 3), "steps": steps,
                         "dtype": "bf16 encoder (f32 accumulate) + f32 GNN" if gnn == "f32" else "bf16 encoder + bf16 GNN Linears (f32 accumulate)",
                         "workload": f"BASELINE.json configs[2]: batch={g2} 8-node fully-connected graphs, 224x224, bf16 activations + MFMA "
                                     f"conv, {args.streams} streams, inputs resident in HBM",
                         "roofline": bf16_roofline(kt)}
        del x2, d2
        # what the bf16 matrix pipe of THIS device sustains with nothing else drawing power (registers-only MFMA stream,
        # rpg_probe_mfma_bf16): the data-sheet 2.5 PFLOP/s holds on zero operands only
        try:
            pipe = {k: round(ops.probe_mfma_bf16(k, iters=20000), 3) for k in ("zeros", "relu_like", "random")}
            out["bf16_matrix_pipe_sustained_pflops"] = dict(pipe, what="chip-wide v_mfma_f32_32x32x16_bf16 on registers only (no LDS / L2 / "
                                                            "HBM traffic), 2 x CUs workgroups x 8 waves x 20000 x 16 MFMAs, HIP events; the clock under the load "
                                                            "depends on the operand data (power cap)")
            for name in ("configs2_bf16_encoder", "configs2_bf16_all"):
                tf = out[name]["roofline"]["achieved"]
                out[name]["roofline"]["frac_of_sustained_pipe_relu_like"] = round(tf / 1e3 / pipe["relu_like"], 4)
                out[name]["roofline"]["frac_of_sustained_pipe_random"] = round(tf / 1e3 / pipe["random"], 4)
        except Exception as e:  # noqa: BLE001 -- a measurement aid: never fail the line over it
            out["bf16_matrix_pipe_sustained_pflops"] = {"error": f"{type(e).__name__}: {e}"}
        model.hip_streams = args.streams
        # evaluation-shape streams: single-graph Data objects in pageable host memory, as a loader delivers them (test.py:193,211)
        h, w, mb = 256, 341, 64
        model.input_img_height = h
        gen = torch.Generator().manual_seed(77)
        ei8 = fc_edge_index(NODES)
        pool = [(torch.randn((NODES, 3 * h * w), generator=gen), torch.randn((NODES, 6), generator=gen) * 0.3) for _ in range(64)]
        # The reference's loader delivers PINNED tensors (DataLoader(..., pin_memory=True), testing/test.py:193): that is the primary
        # stream number; the same stream out of pageable memory (a plain loader) is reported next to it.
        pinned_pool = [(px.pin_memory(), py) for px, py in pool]
        for name, dt_name, n in (("configs3_eval_stream_1gpu_host_fp32", "f32", 2000), ("configs4_eval_stream_1gpu_host_bf16", "bf16", 4000)):
            model.encoder_dtype = model.gnn_dtype = dt_name
            legs = {}
            # (bf16 leg pinned_direct: bf16_input=False = the loader's pinned fp32 tensors sent as they are, rounded on the device --
            # what a rank of an 8-rank host does with its 2 staging threads; `pinned` = the automatic choice of THIS process)
            for leg, src, bfin in (("pinned", pinned_pool, None), ("pageable", pool, None)) + ((("pinned_direct", pinned_pool, False),) if dt_name == "bf16" else ()):
                graphs = [Data(x=src[i % len(src)][0], edge_index=ei8, y=src[i % len(src)][1]) for i in range(n)]
                E.evaluate_stream(model, graphs[:2 * mb], dev, micro_batch=mb, bf16_input=bfin)   # warm-up: packing, workspaces, staging buffers
                torch.cuda.synchronize()
                stats = {}
                t0 = time.perf_counter()
                res = E.evaluate_stream(model, graphs, dev, micro_batch=mb, stats=stats, bf16_input=bfin)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                assert res.pred_poses.shape == (n, 7) and bool((res.pred_poses == res.pred_poses).all())
                legs[leg] = {"value": round(n / dt, 1), "seconds": round(dt, 3), "h2d_gb_per_s": round(stats.get("h2d_bytes", 0) / dt / 1e9, 2),
                             "staged_gb": round(stats.get("staged_bytes", 0) / 1e9, 3), "direct_gb": round(stats.get("direct_bytes", 0) / 1e9, 3),
                             "staging_workers": stats.get("staging_workers")}
                del graphs
            prim = legs["pinned"]
            out[name] = {"value": prim["value"], "unit": "graphs/s", "seconds": prim["seconds"], "graphs": n, "input": "pinned",
                         "dtype": "f32" if dt_name == "f32" else "bf16 encoder + bf16 GNN Linears (f32 accumulate)",
                         "h2d_gb_per_s": prim["h2d_gb_per_s"], "legs": legs,
                         "workload": f"BASELINE.json configs[{3 if dt_name == 'f32' else 4}] shape on ONE GPU: {n} 8-node FC graphs of {h}x{w} synthetic "
                                     f"images in PINNED host memory (the reference's DataLoader(pin_memory=True), test.py:193) -> evaluate_stream "
                                     f"(micro-batch {mb}, H2D on a copy stream"
                                     + (": fp32 sources go straight from the loader's pinned tensors, no staging copy" if dt_name == "f32" else
                                        ": with >= 8 staging threads (this process) the images are rounded to bf16 by the staging threads while the "
                                        "previous micro-batch is post-processed (half the H2D bytes: the link carries 6.1 k graphs/s of fp32 images, "
                                        "the forward does 8 k); leg pinned_direct sends the loader's pinned fp32 tensors as they are (no host pass: "
                                        "what a rank with 2-4 staging threads does)")
                                     + ", D2H + test.py:213-251 post-processing per graph included); legs.pageable = the same stream out of "
                                     "pageable memory; the 4- / 8-GPU sharding is tools/eval_stream.py under torch.distributed.run"}
        del pinned_pool
        # single graph at the evaluation shape (what the unmodified testing/test.py:192-211 loop feeds: batch_size=1, 8 x 256x341)
        model.encoder_dtype = model.gnn_dtype = "f32"
        d1 = fc_batch(torch.randn((NODES, 3 * h * w), generator=torch.Generator(device=dev).manual_seed(5), device=dev), NODES)
        for _ in range(5):
            model(d1)
        torch.cuda.synchronize()
        ts = []
        for _ in range(30):
            t0 = time.perf_counter()
            a1, r1, _ = model(d1)
            r1.cpu()                                              # test.py:214: the caller's own synchronisation point
            ts.append(time.perf_counter() - t0)
        out["latency_1graph_256x341"] = {"ms": round(1e3 * sorted(ts)[len(ts) // 2], 4), "graphs_per_s": round(1.0 / sorted(ts)[len(ts) // 2], 1),
                                         "what": "one 8-node 256x341 graph per forward + .cpu() of the relative poses per call (median of 30): "
                                                 "what INTEGRATION.md's 3-line edit of testing/test.py yields per iteration, fp32"}
        # the reference's loop AS WRITTEN (batch_size = 1 loader, model(data.to(device)), .cpu().data.numpy() per graph:
        # test.py:205-251) over relpose_gnn_amd.lookahead: the loader is read 64 graphs ahead and the forwards are batched
        from relpose_gnn_amd.graph import Batch
        from relpose_gnn_amd.lookahead import lookahead
        items = [Batch.from_data_list([Data(x=px, edge_index=ei8, y=py)]) for px, py in pool]     # what DataLoader(batch_size=1) collates

        class _Loader:
            batch_size = 1

            def __init__(self, n):
                self.n = n

            def __len__(self):
                return self.n

            def __iter__(self):
                return (items[i % len(items)] for i in range(self.n))

        def ref_loop(n):
            loader, wrapped = lookahead(_Loader(n), model, dev, micro_batch=mb)
            preds = []
            for batch_idx, data in enumerate(loader):
                output, output_R, edge_index = wrapped(data.to(dev))
                s = output.size()
                output_R = output_R.cpu().data.numpy().reshape((-1, s[-1]))
                target = data.y.to("cpu").numpy().reshape((-1, s[-1]))
                edges = edge_index.cpu().data.numpy()
                preds.append(E.query_pose(output_R, target, edges, np.zeros(3), np.ones(3), 0)[0])
            return np.stack(preds), wrapped

        ref_loop(2 * mb)
        torch.cuda.synchronize()
        n = 1024
        t0 = time.perf_counter()
        preds, wrapped = ref_loop(n)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert preds.shape == (n, 7) and bool((preds == preds).all()) and wrapped.direct_calls == 0
        out["reference_loop_lookahead_256x341"] = {
            "value": round(n / dt, 1), "unit": "graphs/s", "seconds": round(dt, 3), "graphs": n, "forwards": wrapped.forwards, "dtype": "f32",
            "what": "testing/test.py:205-251 as written (one graph per iteration from a batch_size=1 loader in pageable host memory, "
                    "model(data.to(device)), .cpu().data.numpy(), query pose per graph) with the loader and the module wrapped by "
                    f"relpose_gnn_amd.lookahead.lookahead(loader, model, device, micro_batch={mb}); the same loop on the bare module: "
                    "latency_1graph_256x341"}
    finally:
        model.encoder_dtype, model.gnn_dtype, model.hip_streams, model.input_img_height = "f32", "f32", args.streams, IMG
    return out


def reference_default_flags(model_factory, args, dev, x, base_ms):
    """What the reference's default CLI actually runs (testing/test.py:308-309: --knn 4 --droprate 0.5), timed on the headline
    workload (32 graphs x 8 x 224x224, fp32, inputs resident): `knn=4` builds the kNN graph per stream slot from the slot's
    encoder output, `droprate=0.5` adds the always-on F.dropout + the heads per slot (posenet.py:1047-1048,1073-1086)."""
    from relpose_gnn_amd.graph import fc_batch
    out = {}
    data = fc_batch(x, NODES)
    steps = max(args.steps, 5)
    for name, kw in (("knn4", {"knn": 4, "droprate": 0.0}), ("droprate0.5", {"knn": -1, "droprate": 0.5}),
                     ("knn4_droprate0.5", {"knn": 4, "droprate": 0.5})):
        m = model_factory(**kw)
        m.hip_streams = args.streams
        for _ in range(3):
            _, rel, ei = m(data)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            _, rel, ei = m(data)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert bool(torch.isfinite(rel).all()) and rel.shape[0] == ei.shape[1]
        ms = 1e3 * dt / steps
        out[name] = {"value": round(x.shape[0] // NODES * steps / dt, 1), "unit": "graphs/s", "ms_per_step": round(ms, 3),
                     "edges_per_graph": int(ei.shape[1]) // (x.shape[0] // NODES), "over_headline_step": round(ms / base_ms, 4)}
        del m
    return out


def spawn_ranks(n: int, script: str = None, argv=None) -> int:
    """`python bench.py --gpus N` outside torchrun: run the N ranks as a child `torch.distributed.run` (this process has
    not initialised the GPU: nothing before this point calls into HIP) and relay rank 0's JSON line."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), script or os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv)
    # HSA_ENABLE_IPC_MODE_LEGACY=0: RCCL shares device buffers between the ranks' processes through HIP IPC handles, and the
    # host driver of this GPU pool only supports the dmabuf flavour (with the legacy mode hipIpcGetMemHandle fails with
    # "invalid argument" and init_process_group / the first collective dies).  The image exports it already; it is set
    # here (never overriding a value the caller chose) so that the ranks get it even from a scrubbed environment.
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    dec = json.JSONDecoder()
    for out in proc.stdout:
        # rank 0's line; the ranks share one pipe, so another rank's output may sit in front of / behind it on the same line
        i = out.find('{"metric"')
        obj = None
        if i >= 0:
            try:
                obj, end = dec.raw_decode(out[i:])
            except ValueError:
                obj = None
        if obj is not None:
            line = out[i:i + end]
            rest = (out[:i] + out[i + end:]).strip()
            if rest:
                sys.stderr.write(rest + "\n")
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    return rc if rc else (0 if line is not None else 1)


def main():
    if len(sys.argv) >= 4 and sys.argv[1] == "--cpu-worker":
        return cpu_worker(float(sys.argv[2]), int(sys.argv[3]))
    args = parse()
    if (args.gpus > 1 or args.launcher) and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus, argv=[a for a in sys.argv[1:] if a != "--launcher"]))
    cpu_line = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.cpu_baseline_seconds > 0:
        cpu_line = cpu_baseline(args.cpu_baseline_seconds)      # child processes, before any GPU initialisation here
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP kernels are the only compute path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    # under a launcher (torchrun environment) the process group is RCCL at ANY world size, 1 included: the collective of
    # the step then really runs (`python bench.py --gpus 1 --launcher` is how a one-GPU box covers the N > 1 code path)
    under_launcher = "WORLD_SIZE" in os.environ
    host_cpus = None
    if under_launcher:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        if world > 1:
            # one process per GPU on a shared two-socket host: the launching thread (≈150 kernel launches per step) and whatever
            # it allocates stay on this rank's share of the cores, on its GPU's NUMA node where sysfs names it (plain
            # sched_setaffinity after init: no exec, no numactl hop; RPG_BIND_RANKS=0 switches it off)
            from relpose_gnn_amd.shard import bind_rank_to_host_slice
            host_cpus = bind_rank_to_host_slice(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)), local_rank)

    import relpose_gnn_amd.synth as S
    from relpose_gnn_amd import ops
    from relpose_gnn_amd.graph import fc_batch
    from relpose_gnn_amd.posenet import PoseNetX_R2
    from relpose_gnn_amd.resnet import resnet34
    from relpose_gnn_amd.shard import gather_rows

    D = 2048
    sd = S.synth_state_dict(S.posenet_r2_param_shapes(D, D, D), seed=1)

    def model_factory(knn=-1, droprate=0.0):
        m = PoseNetX_R2(resnet34(), droprate=droprate, pretrained=False, feat_dim=D, edge_feat_dim=D, node_dim=D,
                        input_img_height=IMG, use_gnn=True, knn=knn, use_AP=True, gnn_recursion=2)
        m.load_state_dict(sd)
        return m.to(dev).eval()

    model = model_factory()
    model.hip_streams = args.streams
    if args.schedule:
        model.stream_schedule = [tuple(int(v) for v in part.split(":")) for part in args.schedule.split(",")]
    model.encoder_dtype = args.encoder_dtype
    model.gnn_dtype = args.gnn_dtype
    if args.bf16_bk:
        ops.set_tuning(ops.TUNE_BF16_BK, args.bf16_bk)
    for kv in args.tune:
        k, v = kv.split("=")
        ops.set_tuning(int(k), int(v))

    B = args.graphs
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = torch.randn((NODES * B, 3 * IMG * IMG), generator=gen, device=dev, dtype=torch.float32)
    data = fc_batch(x, NODES)          # edge_index / batch built on x's device
    counts = [B] * world

    def step():
        _, rel, _ = model(data)
        if under_launcher:
            return gather_rows(rel.view(B, NODES * (NODES - 1), 6), counts, always=True)
        return rel

    for _ in range(max(args.warmup, 1)):
        out = step()
    torch.cuda.synchronize()

    def timed(n_steps):
        nonlocal out
        if under_launcher:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            out = step()
        torch.cuda.synchronize()
        if under_launcher:
            dist.barrier()
        return time.perf_counter() - t0

    # pass 1: the K timed steps that define `value` (no instrumentation inside the region)
    elapsed = timed(args.steps)
    # pass 2 (roofline): the same K steps again on ONE stream, every launch of the hot kernels bracketed by HIP events
    # on its launch stream.  One stream, because with two the kernels of the two batch halves overlap and a launch's
    # duration would include sharing the GPU; and outside pass 1, because hipEventRecord costs a few % of the step.
    kt, elapsed_ev = None, None
    if not args.no_kernel_timing:
        model.hip_streams = 1
        step()                         # workspaces for the unsplit batch
        ops.timing_read()              # drop anything recorded so far
        ops.timing_enable(True)
        elapsed_ev = timed(args.steps)
        ops.timing_enable(False)
        kt = ops.timing_read()
        model.hip_streams = args.streams
    model.encoder_dtype = args.encoder_dtype
    if args.bf16_bk:
        ops.set_tuning(ops.TUNE_BF16_BK, args.bf16_bk)
    # the scatter-mean kernel on its own at an HBM-sized launch (2048 graphs: 1.07 GB algorithmic, past L2 + MALL)
    scatter_iso = None
    if kt is not None and rank == 0 and args.encoder_dtype == "f32":
        gi = 2048
        big = fc_batch(torch.empty((NODES * gi, 1), device=dev), NODES)
        gp = ops.graph_prepare(big.edge_index, NODES * gi)
        msg = torch.randn((NODES * (NODES - 1) * gi, D), device=dev)
        for _ in range(2):
            ops.scatter_mean(msg, gp["rowptr"], gp["perm"], NODES * gi)
        ops.timing_read()
        ops.timing_enable(True)
        for _ in range(10):
            ops.scatter_mean(msg, gp["rowptr"], gp["perm"], NODES * gi)
        ops.timing_enable(False)
        v = ops.timing_read()["scatter"]
        gbs = v["work"] / (v["ms"] * 1e-3) / 1e9
        scatter_iso = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": round(gbs / HBM_PEAK_GBS, 4), "launches": v["launches"],
                       "avg_launch_ms": round(v["ms"] / v["launches"], 5),
                       "what": f"the same kernel alone at {gi} graphs per launch ({round(v['work'] / v['launches'] / 1e6)} MB "
                               "algorithmic, back-to-back launches on one stream)"}
        del msg, gp, big
    # single-graph latency (the reference's own evaluation loop is batch_size=1, testing/test.py:192): one 8-node graph per
    # call, host-synchronised after every call (median of 40) and streamed (40 calls, one synchronisation).  Reported next
    # to `value`, never as `value`.
    lat1 = None
    if rank == 0 and args.encoder_dtype == "f32" and not args.no_kernel_timing and not args.no_latency:
        d1 = fc_batch(x[:NODES], NODES)
        for _ in range(5):
            model(d1)
        torch.cuda.synchronize()
        ts = []
        for _ in range(40):
            t0 = time.perf_counter()
            model(d1)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        for _ in range(40):
            model(d1)
        torch.cuda.synchronize()
        lat1 = {"latency_1graph_ms": round(1e3 * sorted(ts)[len(ts) // 2], 4),
                "streamed_1graph_ms": round(1e3 * (time.perf_counter() - t0) / 40, 4),
                "image_streams": int(getattr(model, "small_batch_streams", 1))}
    others = None
    if rank == 0 and world == 1 and args.encoder_dtype == "f32" and args.graphs == 32 and not args.no_other_configs:
        try:
            others = other_configs(model, args, dev)
        except Exception as exc:                                         # a secondary leg must not take the headline down
            others = {"error": f"{type(exc).__name__}: {exc}"}
    ref_flags = None
    if rank == 0 and world == 1 and args.encoder_dtype == "f32" and args.graphs == 32 and not args.no_other_configs:
        try:
            ref_flags = reference_default_flags(model_factory, args, dev, x, 1e3 * elapsed / args.steps)
        except Exception as exc:
            ref_flags = {"error": f"{type(exc).__name__}: {exc}"}
    from relpose_gnn_amd.shard import rank_report
    report = rank_report(dev, elapsed, args.steps, None if host_cpus is None else len(host_cpus))     # collective: every rank
    allgather_ms = None
    if under_launcher:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # the step's one collective on its own: 20 all-gathers of the rel poses back to back (latency-bound: 43 KB per rank)
        rel0 = out[:B].contiguous() if out.shape[0] >= B else out
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            gather_rows(rel0, counts, always=True)
        torch.cuda.synchronize()
        allgather_ms = round(1e3 * (time.perf_counter() - t0) / 20, 4)
    assert out.shape[-1] == 6 and bool(torch.isfinite(out).all())

    if rank == 0:
        graphs = B * world * args.steps
        line = {
            "metric": "graphs/sec (8-node fully-connected, 224x224)", "value": round(graphs / elapsed, 2),
            "unit": "graphs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.encoder_dtype == "f32" else
            ("bf16 encoder (f32 accumulate) + f32 GNN" if args.gnn_dtype == "f32" else "bf16 encoder + bf16 GNN Linears (f32 accumulate)"),
            "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[1]: batch={B} 8-node fully-connected graphs per GPU, 224x224 RGB, "
                                   "fp32, ResNet34 + GNN (D=2048, gnn_recursion=2, droprate=0, eval), random-init weights",
                       "graphs_per_step_per_gpu": B, "nodes_per_graph": NODES, "edges_per_graph": NODES * (NODES - 1),
                       "image": [IMG, IMG], "streams_per_gpu": args.streams,
                       "parallelism": f"graph-sharded x{world}: one process per GPU, RCCL world_size={world}, weights replicated, "
                                      "one all-gather of the rel poses per step (no other collective)",
                       "process_group": (f"nccl (RCCL), world_size={world}: barrier + all_gather_into_tensor per step + all_reduce(MAX) "
                                         "of the elapsed time executed") if under_launcher else
                                        "none (plain single process: no launcher environment, the step ends at the rel poses)"},
        }
        if kt is not None and args.encoder_dtype == "bf16" and kt["conv"]["launches"]:
            line["config"]["workload"] = line["config"]["workload"].replace("configs[1]", "configs[2]").replace(", fp32,", ", bf16 encoder,")
            line["roofline"] = bf16_roofline(kt)
            if kt["linear"]["launches"]:
                v = kt["linear"]
                line["other_kernels"] = {"linear": {"achieved": round(v["work"] / (v["ms"] * 1e-3) / 1e12, 2), "unit": "TFLOP/s",
                                                    "launches": v["launches"], "avg_launch_ms": round(v["ms"] / v["launches"], 5)}}
        elif kt is not None and kt["conv_wino"]["launches"]:
            c = kt["conv_wino"]
            avg_ms = c["ms"] / c["launches"]
            alg = c["work"] / (c["ms"] * 1e-3) / 1e12            # direct-convolution FLOP / time
            exe = c["executed"] / (c["ms"] * 1e-3) / 1e12        # FLOP the matrix pipe issues / time
            line["roofline"] = {
                "bound": "mfma", "achieved": round(exe, 2), "peak": F32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(exe / F32_MATRIX_PEAK_TFLOPS, 4), "traffic": None,
                "what": "achieved = EXECUTED matrix-pipe FLOP of the launches (workgroups x K steps x 48 MFMAs x 8 waves x 4096, "
                        "counted by the launcher; equals SQ_INSTS_VALU_MFMA_F32 x 4096 of the PMC profile) / their summed "
                        "HIP-event durations",
                "kernel": "wino43_conv8p_kernel / wino43_conv8_kernel (3x3/stride-1 conv + BN (+residual) + ReLU as 1-D Winograd "
                          "F(4,3) on v_mfma_f32_32x32x2_f32; 29 of the 36 ResNet34 convolutions: the persistent form where a "
                          "launch has more tiles than CUs, else one workgroup per tile); a launch = the kernel (whole tiles "
                          "and, on layers with a tail, its split-K parts in the same grid) plus wino43_fixup_kernel",
                "launches": c["launches"], "avg_launch_ms": round(avg_ms, 4),
                "executed_gflop_per_launch": round(c["executed"] / c["launches"] / 1e9, 3),
                "algorithmic": {"gflop_per_launch": round(c["work"] / c["launches"] / 1e9, 3), "tflops": round(alg, 2),
                                "over_executed": round(c["work"] / c["executed"], 4),
                                "note": "SURVEY 8(d) counts the direct convolution (2*9*Cin MACs per output); F(4,3) issues "
                                        "about half of them, so the algorithmic rate may exceed the matrix peak"},
                "share_of_instrumented_step_time": round(c["ms"] / (1e3 * elapsed_ev), 4),
                "measured_on": f"{args.steps} further steps of the same workload, one stream, per-launch HIP events: "
                               f"{round(1e3 * elapsed_ev / args.steps, 3)} ms/step, vs "
                               f"{round(1e3 * elapsed / args.steps, 3)} ms/step in the timed region "
                               f"({args.streams} concurrent streams, no events)",
            }
            # PMC counters cannot be read from inside the process: they come from a COMMITTED rocprofv3 --pmc profile of
            # this command, used only if it was taken at this batch size, and labelled with the kernel sources it saw
            from relpose_gnn_amd.build import WINOGRAD_SOURCES, source_digest
            tpath = next((p for p in (os.path.join(ROOT, "profiles", f"r{r}_pmc_wino43.json") for r in (6, 5, 4, 3, 2)) if os.path.exists(p)), None)
            if tpath is not None:
                with open(tpath) as f:
                    tj = json.load(f)
                if tj.get("graphs_per_step") == B:
                    # the profile is stamped with the digest of the files named in its `digest_of` (r3: the Winograd translation
                    # unit + the shared header; r2 profiles: every kernel source)
                    same = tj.get("source_digest") == source_digest(tj.get("digest_of") or (None if "digest_of" not in tj else WINOGRAD_SOURCES))
                    # `traffic` only from a profile of the kernels that are running: otherwise it stays null
                    line["roofline"]["traffic"] = round(tj["traffic_bytes_per_launch"]) if same else None
                    line["roofline"]["committed_profile"] = {
                        "file": "profiles/" + os.path.basename(tpath), "source_digest": tj.get("source_digest"),
                        "git_commit": tj.get("git_commit"), "matches_running_kernels": same,
                        "graphs_per_step": tj.get("graphs_per_step"),
                        "what": "traffic = 1024 * (2 * FETCH_SIZE + WRITE_SIZE) per launch of the main kernel, separate "
                                "--pmc passes, read side doubled per MI355X_MICROARCH.md; NOT observed in this run",
                        **{k: tj[k] for k in ("traffic_bytes_per_launch", "algorithmic_bytes_per_launch", "traffic_over_algorithmic", "mfma_busy_frac",
                                              "cu_busy_frac", "shader_clock_ghz", "executed_mfma_gflop_per_launch") if k in tj}}
            other = {}
            for k in ("conv", "linear", "attention", "scatter", "att_agg"):
                v = kt[k]
                if not v["launches"]:
                    continue
                if k == "att_agg":
                    gbs = v["work"] / (v["ms"] * 1e-3) / 1e9
                    other[k] = {"bound": "valu (exp) / hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(gbs / HBM_PEAK_GBS, 4), "launches": v["launches"],
                                "avg_launch_ms": round(v["ms"] / v["launches"], 5),
                                "what": "attention rows + mean aggregation in one kernel: the in-pipeline scatter-mean launch "
                                        "is gone, its 524,736 algorithmic bytes per graph are part of this kernel's (which also "
                                        "reads the attention operands, 172,032 B per graph, and evaluates 65,536 exp per edge: "
                                        "that, not HBM, bounds it: 117 M exp + 352 M FMA per launch at 32 graphs)"}
                    continue
                if k == "scatter":
                    gbs = v["work"] / (v["ms"] * 1e-3) / 1e9
                    other[k] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(gbs / HBM_PEAK_GBS, 4), "launches": v["launches"],
                                "avg_launch_ms": round(v["ms"] / v["launches"], 5),
                                "what": "in-pipeline launches (524,736 algorithmic bytes per graph per launch)"}
                else:
                    tf = v["executed"] / (v["ms"] * 1e-3) / 1e12
                    other[k] = {"achieved": round(tf, 2), "unit": "TFLOP/s", "frac": round(tf / F32_MATRIX_PEAK_TFLOPS, 4),
                                "algorithmic_tflops": round(v["work"] / (v["ms"] * 1e-3) / 1e12, 2),
                                "launches": v["launches"], "avg_launch_ms": round(v["ms"] / v["launches"], 5),
                                "share_of_instrumented_step_time": round(v["ms"] / (1e3 * elapsed_ev), 4)}
            if "conv" in other:
                other["conv"]["kernel"] = "direct implicit-GEMM conv (stem 7x7/2, three 3x3/2, three 1x1/2)"
            if scatter_iso is not None:
                other["scatter_isolated"] = scatter_iso
            line["other_kernels"] = other
        if lat1 is not None:
            line["latency_1graph"] = dict(lat1, what="one 8-node 224x224 graph per forward (the reference's batch_size=1 loop): wall time "
                                                    "per call with a host synchronisation after each / per call when 40 calls are streamed")
        if others is not None:
            line["other_configs"] = others
            # FLAT scalar copies inside `config` (the driver's record keeps the scalar values of `config`, not nested objects:
            # VERDICT r5 item 2); the nested forms stay under `other_configs` / `reference_default_flags`
            cfg = line["config"]
            for short, key in (("c2_bf16enc", "configs2_bf16_encoder"), ("c2_bf16all", "configs2_bf16_all")):
                v = others.get(key)
                if isinstance(v, dict):
                    cfg[short + "_gps"] = v.get("value")
                    cfg[short + "_ms"] = v.get("ms_per_step")
                    cfg[short + "_frac"] = (v.get("roofline") or {}).get("frac")
                    if (v.get("roofline") or {}).get("frac_of_sustained_pipe_relu_like") is not None:
                        cfg[short + "_frac_of_sustained_pipe"] = v["roofline"]["frac_of_sustained_pipe_relu_like"]
            pipe = others.get("bf16_matrix_pipe_sustained_pflops")
            if isinstance(pipe, dict) and "relu_like" in pipe:          # measured ceiling of the bf16 matrix pipe on this box, by operand data
                cfg["bf16_pipe_zeros_pflops"], cfg["bf16_pipe_relu_like_pflops"], cfg["bf16_pipe_random_pflops"] = pipe["zeros"], pipe["relu_like"], pipe["random"]
            for short, key in (("c3_stream_fp32", "configs3_eval_stream_1gpu_host_fp32"), ("c4_stream_bf16", "configs4_eval_stream_1gpu_host_bf16")):
                v = others.get(key)
                if isinstance(v, dict):
                    cfg[short + "_gps"] = v.get("value")                    # pinned sources (the reference's loader)
                    for leg, lv in (v.get("legs") or {}).items():
                        if leg != "pinned":
                            cfg[f"{short}_{leg}_gps"] = lv.get("value")
                    cfg[short + "_pinned_staged_gb"] = ((v.get("legs") or {}).get("pinned") or {}).get("staged_gb")
            if isinstance(others.get("latency_1graph_256x341"), dict):
                cfg["lat1_256x341_ms"] = others["latency_1graph_256x341"]["ms"]
            if isinstance(others.get("reference_loop_lookahead_256x341"), dict):       # test.py's own loop over lookahead(): graphs/s
                cfg["c3_ref_loop_lookahead_gps"] = others["reference_loop_lookahead_256x341"]["value"]
            if "error" in others:
                cfg["secondary_error"] = str(others["error"])[:120]
        if lat1 is not None:
            line["config"]["lat1_ms"] = lat1["latency_1graph_ms"]
        if isinstance(ref_flags, dict):
            for kk, v in ref_flags.items():
                if isinstance(v, dict):
                    line["config"]["c1_" + kk.replace(".", "") + "_gps"] = v.get("value")
                else:
                    line["config"]["c1_flags_error"] = str(v)[:120]
        if ref_flags is not None:
            line["reference_default_flags"] = ref_flags
        import hashlib
        from relpose_gnn_amd import _lib as _L
        with open(_L.LIB_PATH, "rb") as fh:
            lib_sha = hashlib.sha256(fh.read()).hexdigest()[:12]
        lib_rel = os.path.relpath(_L.LIB_PATH, ROOT) if _L.LIB_PATH.startswith(ROOT) else _L.LIB_PATH
        line["config"]["loaded_library"] = f"{lib_rel}@{lib_sha}" + (" (RPG_HIP_LIB override)" if os.environ.get("RPG_HIP_LIB") else "")
        # what the collectives themselves observed (shard.rank_report): flat scalars, so a < 7x result at 8 GPUs can be attributed
        # to a rank (slowest_rank, rank_ms_spread), a socket (numa_nodes_seen, host_cpus_*), a launcher mistake (distinct_gpus <
        # ranks) or the collective (allgather_ms against ms_per_step)
        line["rccl_ranks_seen"] = report["rccl_ranks_seen"]
        for kk, v in report.items():
            line["config"][kk] = v
        if allgather_ms is not None:
            line["config"]["allgather_ms"] = allgather_ms
        if cpu_line is not None:
            line["cpu_baseline"] = cpu_line
        print(json.dumps(line), flush=True)
    if under_launcher:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
