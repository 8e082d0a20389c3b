"""Round-6 probe: the bf16 evaluation stream (BASELINE configs[4] shape on one GPU: 8 x 256x341 per graph, micro-batches of 64) out of
pinned / pageable host memory, with and without the host-side bf16 rounding pass, and the rounding helper's own rate per instruction set
and thread count.   python tools/probes/c4_legs.py [graphs]"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import test_hip_history as T  # noqa: E402
from relpose_gnn_amd import _lib  # noqa: E402
from relpose_gnn_amd import evaluate as E  # noqa: E402
from relpose_gnn_amd.graph import Data, fc_edge_index  # noqa: E402

lib = _lib.lib()
n = 32 * 1024 * 1024
src = [np.random.randn(n).astype(np.float32) for _ in range(32)]
dst = [np.empty(n, dtype=np.uint16) for _ in range(32)]
for isa in (1, 2, 3):
    for threads in (1, 8, 16, 32):
        def work(i):
            return lib.rpg_host_f32_to_bf16_isa(src[i].ctypes.data, dst[i].ctypes.data, n, isa)
        with ThreadPoolExecutor(threads) as pool:
            list(pool.map(work, range(threads)))
            t0 = time.perf_counter()
            rc = list(pool.map(work, range(threads)))
            dt = time.perf_counter() - t0
        name = ("scalar", "AVX2", "AVX-512F")[isa - 1]
        print(f"rpg_host_f32_to_bf16 isa {isa} ({name}) threads {threads:2d}: rc {rc[0]}  {threads * n * 4 / dt / 1e9:7.1f} GB/s of fp32", flush=True)
del src, dst

dev = torch.device("cuda:0")
h, w, mb = 256, 341, 64
n_graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
m = T._model(dev, h)
m.encoder_dtype = m.gnn_dtype = "bf16"
gen = torch.Generator().manual_seed(77)
ei8 = fc_edge_index(8)
pool = [(torch.randn((8, 3 * h * w), generator=gen), torch.randn((8, 6), generator=gen) * 0.3) for _ in range(64)]
pinned = [(px.pin_memory(), py) for px, py in pool]
for rep in range(2):
    for leg, srcs, bfin in (("pinned, auto", pinned, None), ("pinned, host-rounded", pinned, True), ("pageable (host-rounded)", pool, None)):
        graphs = [Data(x=srcs[i % 64][0], edge_index=ei8, y=srcs[i % 64][1]) for i in range(n_graphs)]
        E.evaluate_stream(m, graphs[:2 * mb], dev, micro_batch=mb, bf16_input=bfin)
        torch.cuda.synchronize()
        stats = {}
        t0 = time.perf_counter()
        E.evaluate_stream(m, graphs, dev, micro_batch=mb, stats=stats, bf16_input=bfin)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"c4 {leg:26s}: {n_graphs / dt:7.1f} graphs/s  h2d {stats.get('h2d_bytes', 0) / dt / 1e9:5.1f} GB/s  staged {stats.get('staged_bytes', 0) / 1e9:6.2f} GB  "
              f"direct {stats.get('direct_bytes', 0) / 1e9:6.2f} GB  workers {stats.get('staging_workers')}", flush=True)
