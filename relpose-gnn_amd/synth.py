"""Deterministic synthetic weights and inputs for the PoseNetX_R2 hot path.

There is no network for checkpoints or datasets, so tests and ``bench.py`` use random-init
weights of the reference architecture.  To let the GPU box regenerate *bit-identical* weights
without shipping 300 MB of fixtures, values come from a counter-based integer hash evaluated
with numpy uint64 arithmetic (no transcendental functions, so the result does not depend on
the libm / SIMD dispatch of the host).

State-dict key names and shapes follow the reference model (SURVEY.md section 8(b);
/root/reference/python/niantic/modules/posenet.py:923-997, my_gnn_layer.py:277-291,
att.py:8-14, torchvision 0.9.1 resnet34).
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np
import torch

RESNET34_BLOCKS = (3, 4, 6, 3)
RESNET34_PLANES = (64, 128, 256, 512)

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _mix(z: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def hash_uniform(name: str, numel: int, seed: int = 0, lane: int = 0) -> np.ndarray:
    """numel float64 values in [0,1) with 24 random bits each, a pure function of
    (name, seed, lane, element index)."""
    key = np.uint64((zlib.crc32(name.encode()) << 20) ^ (seed * 0x9E3779B1 & 0xFFFFFFFF) ^ (lane << 56))
    with np.errstate(over="ignore"):
        ctr = np.arange(numel, dtype=np.uint64) * _GOLD + key
        bits = _mix(_mix(ctr) + key)
    return (bits >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))


def hash_normal(name: str, shape, std: float = 1.0, mean: float = 0.0, seed: int = 0) -> torch.Tensor:
    """Approximately normal fp32 tensor (Irwin-Hall sum of 4 hashed uniforms, unit variance)."""
    n = int(np.prod(shape)) if len(shape) else 1
    s = np.zeros(n, dtype=np.float64)
    for lane in range(4):
        s += hash_uniform(name, n, seed, lane)
    s = (s - 2.0) * np.sqrt(3.0) * std + mean      # var(sum of 4 U) = 4/12
    return torch.from_numpy(s.astype(np.float32).reshape(shape))


def hash_range(name: str, shape, lo: float, hi: float, seed: int = 0) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    u = hash_uniform(name, n, seed, 7)
    return torch.from_numpy((lo + (hi - lo) * u).astype(np.float32).reshape(shape))


# --------------------------------------------------------------------------- #
# parameter inventory
# --------------------------------------------------------------------------- #
def resnet34_param_shapes(prefix: str = "feature_extractor.", feat_dim: int = 2048,
                          planes=RESNET34_PLANES, blocks=RESNET34_BLOCKS) -> "OrderedDict[str, Tuple[int, ...]]":
    """Key -> shape of the torchvision-style ResNet (BasicBlock) state dict, fc replaced by
    Linear(512, feat_dim) as posenet.py:942-945 does."""
    sd: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()

    def bn(p, c):
        sd[p + "weight"] = (c,)
        sd[p + "bias"] = (c,)
        sd[p + "running_mean"] = (c,)
        sd[p + "running_var"] = (c,)
        sd[p + "num_batches_tracked"] = ()

    sd[prefix + "conv1.weight"] = (planes[0], 3, 7, 7)
    bn(prefix + "bn1.", planes[0])
    cin = planes[0]
    for li, (c, nb) in enumerate(zip(planes, blocks), start=1):
        for bi in range(nb):
            p = f"{prefix}layer{li}.{bi}."
            stride = 2 if (li > 1 and bi == 0) else 1
            sd[p + "conv1.weight"] = (c, cin, 3, 3)
            bn(p + "bn1.", c)
            sd[p + "conv2.weight"] = (c, c, 3, 3)
            bn(p + "bn2.", c)
            if stride != 1 or cin != c:
                sd[p + "downsample.0.weight"] = (c, cin, 1, 1)
                bn(p + "downsample.1.", c)
            cin = c
    sd[prefix + "fc.weight"] = (feat_dim, planes[-1])
    sd[prefix + "fc.bias"] = (feat_dim,)
    return sd


def posenet_r2_param_shapes(feat_dim: int = 2048, edge_feat_dim: int = 2048, node_dim: int = 2048,
                            planes=RESNET34_PLANES, blocks=RESNET34_BLOCKS, use_attention: bool = False,
                            use_AP: bool = True, L: int = 1) -> "OrderedDict[str, Tuple[int, ...]]":
    """Full PoseNetX_R2 (use_gnn=True) state dict inventory, in the order the reference module registers its
    children (posenet.py:941-975): encoder, proj_edge, gnn1..gnnL, [att], fc_xyz, fc_wpqr, fc_xyz_R, fc_wpqr_R."""
    assert feat_dim == node_dim, "the reference feeds encoder features straight into gnn1 (posenet.py:1063)"
    sd = resnet34_param_shapes("feature_extractor.", feat_dim, planes, blocks)
    D, De = node_dim, edge_feat_dim
    sd["proj_edge.weight"] = (De, 2 * feat_dim)
    sd["proj_edge.bias"] = (De,)
    # simpleConvEdge_upt(node_dim, edge_feat_dim, node_dim): mlp, mlp_updating, edge_model, att
    for li in range(1, L + 1):
        g = f"gnn{li}."
        sd[g + "mlp.0.weight"] = (D, D + De)
        sd[g + "mlp.0.bias"] = (D,)
        sd[g + "mlp.2.weight"] = (D, D)
        sd[g + "mlp.2.bias"] = (D,)
        sd[g + "mlp_updating.0.weight"] = (D, 2 * D)
        sd[g + "mlp_updating.0.bias"] = (D,)
        sd[g + "mlp_updating.2.weight"] = (D, D)
        sd[g + "mlp_updating.2.bias"] = (D,)
        sd[g + "edge_model.edge_mlp.0.weight"] = (De, 2 * D + De)
        sd[g + "edge_model.edge_mlp.0.bias"] = (De,)
        sd[g + "edge_model.edge_mlp.2.weight"] = (De, De)
        sd[g + "edge_model.edge_mlp.2.bias"] = (De,)
        for n in ("g", "theta", "phi"):
            sd[f"{g}att.{n}.weight"] = (D // 8, D)
            sd[f"{g}att.{n}.bias"] = (D // 8,)
        sd[g + "att.W.weight"] = (D, D // 8)
        sd[g + "att.W.bias"] = (D,)
    if use_attention:                                   # self.att = AttentionBlock(feat_dim), posenet.py:961-962
        for n in ("g", "theta", "phi"):
            sd[f"att.{n}.weight"] = (feat_dim // 8, feat_dim)
            sd[f"att.{n}.bias"] = (feat_dim // 8,)
        sd["att.W.weight"] = (feat_dim, feat_dim // 8)
        sd["att.W.bias"] = (feat_dim,)
    for n in ("fc_xyz", "fc_wpqr"):
        sd[n + ".weight"] = (3, D if use_AP else 2 * D)
        sd[n + ".bias"] = (3,)
    for n in ("fc_xyz_R", "fc_wpqr_R"):
        sd[n + ".weight"] = (3, De)
        sd[n + ".bias"] = (3,)
    return sd


def synth_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """Random-init values for every key in ``shapes``.

    Conv / Linear weights are zero-mean with variance 2/fan_in (convs, first Linear of a
    ReLU MLP) or 1/fan_in (other Linears); biases N(0, 0.05).  BatchNorm statistics are
    randomised so BN is not an identity: running_var in [0.5,1.5], running_mean N(0,0.1),
    gamma in [0.5,1.5] (in [0.2,0.6] for the last BN of a residual branch so 16 residual
    adds do not blow activations up), beta N(0,0.1).
    """
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k, shp in shapes.items():
        if k.endswith("num_batches_tracked"):
            out[k] = torch.tensor(0, dtype=torch.int64)
        elif k.endswith("running_mean"):
            out[k] = hash_normal(k, shp, 0.1, 0.0, seed)
        elif k.endswith("running_var"):
            out[k] = hash_range(k, shp, 0.5, 1.5, seed)
        elif ".bn" in k or "downsample.1." in k:
            if k.endswith("weight"):
                lo, hi = (0.2, 0.6) if ".bn2." in k else (0.5, 1.5)
                out[k] = hash_range(k, shp, lo, hi, seed)
            else:
                out[k] = hash_normal(k, shp, 0.1, 0.0, seed)
        elif k.endswith("bias"):
            out[k] = hash_normal(k, shp, 0.05, 0.0, seed)
        else:
            fan_in = int(np.prod(shp[1:]))
            relu_fed = len(shp) == 4 or k.endswith(".0.weight") or k.startswith("proj_edge")
            out[k] = hash_normal(k, shp, float(np.sqrt((2.0 if relu_fed else 1.0) / fan_in)), 0.0, seed)
    return out


def synth_images(n_nodes: int, h: int, w: int, seed: int = 0) -> torch.Tensor:
    """data.x: [n_nodes, 3*h*w] fp32, ~N(0,1) (normalised image statistics)."""
    return hash_normal("data.x", (n_nodes, 3 * h * w), 1.0, 0.0, seed)
