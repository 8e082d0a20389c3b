"""Round 5: rounding noise of the fp32 GNN Linears against float64 as a function of RPG_TUNE_FOLD_K (two-level accumulation
of the v_mfma_f32_32x32x2_f32 chains).  16 graphs x 8 x 256x341 iid-noise images (the configs[3] geometry whose abs poses
sat at 1.0e-4 of the CPU fp32 oracle in round 4); HIP encoder features -> {HIP GNN at each fold_k, oracle GNN fp32, oracle GNN
fp64} on the SAME features.  Prints max-norm relative errors of the abs / rel poses against the float64 answer.

    gpurun -- python tools/probes/fold_probe.py [graphs]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import relpose_gnn_amd.synth as S  # noqa: E402
from oracle import posenet_ref as O  # noqa: E402  (checker)
from relpose_gnn_amd import ops  # noqa: E402
from relpose_gnn_amd.graph import fc_batch  # noqa: E402
from relpose_gnn_amd.posenet import PoseNetX_R2  # noqa: E402
from relpose_gnn_amd.resnet import resnet34  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H, W, D = 256, 341, 2048
dev = torch.device("cuda:0")
m = PoseNetX_R2(resnet34(), droprate=0.0, pretrained=False, feat_dim=D, edge_feat_dim=D, node_dim=D, input_img_height=H,
                use_gnn=True, knn=-1, use_AP=True, gnn_recursion=2)
sd = S.synth_state_dict(S.posenet_r2_param_shapes(D, D, D), seed=1)
m.load_state_dict(sd)
m = m.to(dev).eval()
m.hip_streams = 1
x = torch.randn((G * 8, 3 * H * W), generator=torch.Generator().manual_seed(8642))
data = fc_batch(x.to(dev), 8)
rel_err = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max())
torch.set_num_threads(min(32, torch.get_num_threads()))
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
ei = O.batch_edge_index(8, G)
out = []
for fold in (0, 1024, 512, 256, 128, 64):
    ops.set_tuning(ops.TUNE_FOLD_K, fold)
    feat = m._enc.run(m.feature_extractor.state_dict, "", data.x.view(-1, 3, H, W)).cpu()
    a, r, _ = m(data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        m(data)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    a64, r64 = O.gnn_forward(sd64, feat.double(), ei, 2)
    a32, r32 = O.gnn_forward(sd, feat, ei, 2)
    rec = {"fold_k": fold, "hip_abs_vs_fp64": rel_err(a.cpu(), a64), "cpu_abs_vs_fp64": rel_err(a32, a64),
           "hip_rel_vs_fp64": rel_err(r.cpu(), r64), "cpu_rel_vs_fp64": rel_err(r32, r64),
           "hip_abs_vs_cpu32": rel_err(a.cpu(), a32), "ms_forward": round(ms, 3)}
    rec["ratio_abs"] = rec["hip_abs_vs_fp64"] / rec["cpu_abs_vs_fp64"]
    print(json.dumps(rec), flush=True)
    out.append(rec)
d = os.path.join(ROOT, "gpurun_out")
os.makedirs(d, exist_ok=True)
with open(os.path.join(d, "fold_probe.jsonl"), "a") as f:
    for rec in out:
        f.write(json.dumps(rec) + "\n")
