"""One 8-node graph per forward (the reference's batch_size = 1 loop, testing/test.py:192-211), 50 forwards: run under
rocprofv3 --kernel-trace --stats for the per-kernel split of the single-graph latency.   [h w] [dtype]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import test_hip_history as T  # noqa: E402
from relpose_gnn_amd.graph import fc_batch  # noqa: E402

h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (224, 224)
dev = torch.device("cuda:0")
m = T._model(dev, h)
m.encoder_dtype = m.gnn_dtype = sys.argv[3] if len(sys.argv) > 3 else "f32"
d = fc_batch(torch.randn((8, 3 * h * w), device=dev), 8)
for _ in range(10):
    m(d)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(50):
    m(d)
torch.cuda.synchronize()
print(f"{h}x{w}: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per forward (streamed)")
