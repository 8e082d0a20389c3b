"""Round-6 probe: launches far beyond the benched sizes (one stream): does the forward still (a) run or refuse loudly, (b) not depend on
its workspaces' content, (c) agree with the same graphs run 32 at a time?   python tools/probes/big_launch.py [graphs ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import test_hip_history as T  # noqa: E402
from relpose_gnn_amd.graph import fc_batch  # noqa: E402

dev = torch.device("cuda:0")
cases = [(tuple(int(v) for v in a.split("x")) if "x" in a else (224, 224, int(a))) for a in sys.argv[1:]] or \
    [(224, 224, 128), (224, 224, 456), (256, 341, 112), (256, 341, 264)]              # arguments: graphs (at 224x224) or HxWxgraphs
for h, w, graphs in cases:
    ref = None
    for dtype in ("f32", "bf16"):
        m = T._model(dev, h)
        m.encoder_dtype = m.gnn_dtype = dtype
        m.hip_streams = 1
        g = torch.Generator(device=dev).manual_seed(9)
        x = torch.randn((graphs * 8, 3 * h * w), generator=g, device=dev)
        try:
            t0 = time.perf_counter()
            a0, r0, _ = m(fc_batch(x, 8))
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for t in m._ws_pool._buf.values():
                t.fill_(255)
            a1, r1, _ = m(fc_batch(x, 8))
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            print(f"{h}x{w} graphs={graphs} {dtype}: refused: {type(e).__name__}: {str(e)[:200]}", flush=True)
            continue
        same = bool(torch.equal(a0, a1) and torch.equal(r0, r1))
        ws = sum(t.numel() for t in m._ws_pool._buf.values()) / 2 ** 30
        m._ws_pool.clear()
        pa, pr = [], []
        for g0 in range(0, graphs, 32):
            g1 = min(graphs, g0 + 32)
            a, r, _ = m(fc_batch(x[g0 * 8:g1 * 8], 8))
            pa.append(a.clone()); pr.append(r.clone())
        pa, pr = torch.cat(pa), torch.cat(pr)
        dg = sorted(set((((a0 != pa).view(graphs, -1).any(dim=1)) | ((r0 != pr).view(graphs, -1).any(dim=1))).nonzero().flatten().tolist()))
        print(f"    graphs whose poses differ bitwise from the pieces: {len(dg)}: {dg[:6]} .. {dg[-6:]}", flush=True)
        ea = float((a0 - pa).abs().max() / pa.abs().max()); er = float((r0 - pr).abs().max() / pr.abs().max())
        print(f"{h}x{w} graphs={graphs} {dtype}: {1e3 * (t1 - t0):.0f} ms first forward, workspace {ws:.1f} GiB, finite {bool(torch.isfinite(a0).all())}, "
              f"poison-independent {same}, vs 32-graph pieces: abs {ea:.2e} rel {er:.2e}", flush=True)
        if dtype == "f32":
            ref = (pa, pr)
        else:
            # bf16 against the fp32 poses of the same graphs, per graph (max-norm relative to the graph's largest fp32 value): the big
            # launch and the 32-graph pieces must sit at the same distance
            def per_graph(u, v, rows):
                uu, vv = u.view(graphs, rows, -1), v.view(graphs, rows, -1)
                return ((uu - vv).abs().amax(dim=(1, 2)) / vv.abs().amax(dim=(1, 2)))
            for name, big, pc, rf, rows in (("abs", a0, pa, ref[0], 8), ("rel", r0, pr, ref[1], 56)):
                eb, ep = per_graph(big, rf, rows), per_graph(pc, rf, rows)
                print(f"    bf16 vs fp32 per graph, {name}: big launch max {float(eb.max()):.3e} mean {float(eb.mean()):.3e} (worst graph {int(eb.argmax())}) | "
                      f"pieces max {float(ep.max()):.3e} mean {float(ep.mean()):.3e}", flush=True)
        del m, x
        torch.cuda.empty_cache()
