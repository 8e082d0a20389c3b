"""Packing of a PoseNetX_R2 state dict into the device tensors the C ABI consumes.

Done once per weight load (``PoseNetX_R2._packed()``), on the model's device, with torch used only to
permute / concatenate / fold constants:

encoder (``rpg_resnet_forward_f32`` tensor order)
    for the stem, then every BasicBlock's conv1, conv2 and (if present) downsample conv, four slots each:
        w_ohwi  [Cout][KH][KW][Cin]   = PyTorch OIHW weight permuted to channels-last; the stem's 3 input
                                        channels are zero-padded to 4 so every im2col chunk is one 16-byte load
        scale   [Cout] = gamma / sqrt(running_var + eps)          (eval-mode BatchNorm folded to an affine
        shift   [Cout] = beta - running_mean * scale               epilogue; the conv weights are untouched)
        u       [6][Cout][3][Cin]     = Winograd F(4,3) weights of a 3x3 stride-1 convolution, or None
                                        (stem: wpack [74][2][64] of the fused stem kernel, see pack_stem_pairs)
    then fc.weight [feat][512], fc.bias [feat]

GNN (``rpg_gnn_forward_f32`` tensor order, 22 tensors + 4 optional split matrices, see pack_gnn)
    proj_edge.{weight,bias}; gnn1.edge_model.edge_mlp.{0,2}.{weight,bias}; gnn1.mlp.{0,2}.{weight,bias};
    att g|theta|phi weights concatenated to [3C][D] and biases to [3C]; gnn1.att.W.{weight,bias};
    gnn1.mlp_updating.{0,2}.{weight,bias}; node heads cat(fc_xyz, fc_wpqr) -> [6][D],[6];
    edge heads cat(fc_xyz_R, fc_wpqr_R) -> [6][D],[6].
Linear weights keep the PyTorch [out][in] layout (both GEMM operands are K-contiguous).

Reference: /root/reference/python/niantic/modules/posenet.py:941-975 (module inventory),
my_gnn_layer.py:280-291, att.py:9-14; BatchNorm eps = 1e-5 (torch default used by torchvision).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch

BN_EPS = 1e-5


def _bn_affine(sd: Dict[str, torch.Tensor], p: str, eps: float = BN_EPS) -> Tuple[torch.Tensor, torch.Tensor]:
    scale = sd[p + "weight"].float() / torch.sqrt(sd[p + "running_var"].float() + eps)
    shift = sd[p + "bias"].float() - sd[p + "running_mean"].float() * scale
    return scale.contiguous(), shift.contiguous()


def _ohwi(w: torch.Tensor, pad_cin_to: int = 0) -> torch.Tensor:
    w = w.float().permute(0, 2, 3, 1)
    if pad_cin_to and w.shape[-1] < pad_cin_to:
        w = torch.nn.functional.pad(w, (0, pad_cin_to - w.shape[-1]))
    return w.contiguous()


def stem_pair_table() -> Tuple[List[Tuple[int, int, int]], List[Optional[Tuple[int, int, int]]]]:
    """The 74 tap pairs (c, kh, kw) of the fused stem kernel (csrc/stem.hip pair_off / pair_class; a host test compares this
    with rpg_stem_pair_table): 63 pairs of neighbouring columns (kw = 0|1, 2|3, 4|5 of every (c, kh)), 9 pairs of
    neighbouring rows in the last column (kh = 0|1, 2|3, 4|5 at kw = 6), the pair (c = 0 | 1) at (6, 6), and (2, 6, 6) alone."""
    a: List[Tuple[int, int, int]] = []
    b: List[Optional[Tuple[int, int, int]]] = []
    for c in range(3):
        for kh in range(7):
            for q in range(3):
                a.append((c, kh, 2 * q)); b.append((c, kh, 2 * q + 1))
    for c in range(3):
        for q in range(3):
            a.append((c, 2 * q, 6)); b.append((c, 2 * q + 1, 6))
    a.append((0, 6, 6)); b.append((1, 6, 6))
    a.append((2, 6, 6)); b.append(None)
    return a, b


def pack_stem_pairs(w_oihw: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """conv1.weight [64][3][7][7] and the folded BatchNorm scale [64] -> wpack of rpg_stem_conv7x7s2_bn_relu_maxpool_f32, two operand
    images back to back ([74][2][64] for the tile kernel, then [2][75][64] for the strip-march kernel): element [kp][nf][l] = scale[ch] * W[ch][tap], ch = 32 nf + (l & 31), tap = first
    (l < 32) or second (l >= 32) tap of pair kp (0 for the missing partner of the last pair)."""
    if tuple(w_oihw.shape) != (64, 3, 7, 7):
        raise ValueError("the fused stem kernel is for Conv2d(3, 64, 7)")
    w = w_oihw.float() * scale.float().view(64, 1, 1, 1)
    a, b = stem_pair_table()
    wa = torch.stack([w[:, c, kh, kw] for c, kh, kw in a])                                   # [74][64]
    wb = torch.stack([w[:, t[0], t[1], t[2]] if t is not None else torch.zeros_like(w[:, 0, 0, 0]) for t in b])
    out = torch.stack([wa.view(74, 2, 32), wb.view(74, 2, 32)], dim=2)                       # [74][nf][half][32]
    tile = out.reshape(74 * 2 * 64)
    # second part (round 6, the strip-march kernel stem_strip_f32_kernel): [nf][75][l = 32 half + n]: per kernel row kh the three
    # horizontal pairs (kw 2p | 2p + 1) of the three channels, after an odd kh the vertical pairs (kh - 1 | kh, kw = 6), after kh = 6
    # the lone (6, 6) taps -- the order the kernel issues its MFMAs in
    zero = torch.zeros_like(w[:, 0, 0, 0])
    sa, sb = [], []
    for kh in range(7):
        for c in range(3):
            for p_ in range(3):
                sa.append(w[:, c, kh, 2 * p_]); sb.append(w[:, c, kh, 2 * p_ + 1])
        if kh & 1:
            for c in range(3):
                sa.append(w[:, c, kh - 1, 6]); sb.append(w[:, c, kh, 6])
        if kh == 6:
            for c in range(3):
                sa.append(w[:, c, 6, 6]); sb.append(zero)
    assert len(sa) == 75
    sa, sb = torch.stack(sa), torch.stack(sb)                                                 # [75][64 ch]
    strip = torch.stack([sa.view(75, 2, 32), sb.view(75, 2, 32)], dim=2)                      # [75][nf][half][32]
    strip = strip.permute(1, 0, 2, 3).reshape(2 * 75 * 64)                                    # [nf][75][half][32]
    return torch.cat([tile, strip]).contiguous()


def resnet_structure(sd: Dict[str, torch.Tensor], prefix: str) -> Tuple[List[int], List[int]]:
    """(blocks per layer, planes per layer) read off the state-dict keys."""
    blocks, planes = [], []
    for li in range(1, 5):
        b = 0
        while f"{prefix}layer{li}.{b}.conv1.weight" in sd:
            b += 1
        if b == 0:
            raise KeyError(f"{prefix}layer{li}.0.conv1.weight missing: not a torchvision-style ResNet state dict")
        blocks.append(b)
        planes.append(int(sd[f"{prefix}layer{li}.0.conv1.weight"].shape[0]))
    return blocks, planes


def pack_resnet(sd: Dict[str, torch.Tensor], prefix: str = "feature_extractor.", wino_fn=None
                ) -> Tuple[List[Optional[torch.Tensor]], List[int], List[int]]:
    """``wino_fn(w_ohwi) -> U [6][Cout][3][Cin]`` (``ops.wino43_transform_weights``, needs the GPU) fills the fourth slot
    of every 3x3 stride-1 convolution; without it (host-only packing) the slot stays None = direct kernel."""
    blocks, planes = resnet_structure(sd, prefix)
    t: List[Optional[torch.Tensor]] = []

    def wino(w_ohwi, stride):
        ok = wino_fn is not None and stride == 1 and w_ohwi.shape[3] % 4 == 0 and w_ohwi.shape[0] % 4 == 0
        return wino_fn(w_ohwi) if ok else None
    w = sd[prefix + "conv1.weight"]
    if tuple(w.shape[1:]) != (3, 7, 7) or w.shape[0] != planes[0]:
        raise ValueError("stem must be Conv2d(3, planes[0], 7, stride 2, pad 3)")
    sc1, sh1 = _bn_affine(sd, prefix + "bn1.")
    # fourth slot: operands of the fused conv + BN + ReLU + max-pool stem kernel (64-channel stems only)
    t += [_ohwi(w, pad_cin_to=4), sc1, sh1, pack_stem_pairs(w, sc1) if w.shape[0] == 64 else None]
    cin = planes[0]
    for li, (nb, c) in enumerate(zip(blocks, planes), start=1):
        for b in range(nb):
            p = f"{prefix}layer{li}.{b}."
            stride = 2 if (li > 1 and b == 0) else 1
            if tuple(sd[p + "conv1.weight"].shape) != (c, cin, 3, 3) or tuple(sd[p + "conv2.weight"].shape) != (c, c, 3, 3):
                raise ValueError(f"{p}: not a BasicBlock with 3x3 convolutions")
            w1, w2 = _ohwi(sd[p + "conv1.weight"]), _ohwi(sd[p + "conv2.weight"])
            t += [w1, *_bn_affine(sd, p + "bn1."), wino(w1, stride)]
            t += [w2, *_bn_affine(sd, p + "bn2."), wino(w2, 1)]
            has_ds = (p + "downsample.0.weight") in sd
            if has_ds != (stride != 1 or cin != c):
                raise ValueError(f"{p}: downsample presence does not match the torchvision BasicBlock rule")
            if has_ds:
                t += [_ohwi(sd[p + "downsample.0.weight"]), *_bn_affine(sd, p + "downsample.1."), None]
            cin = c
    t += [sd[prefix + "fc.weight"].float().contiguous(), sd[prefix + "fc.bias"].float().contiguous()]
    return t, blocks, planes


def pack_stem_bf16(w_oihw: torch.Tensor) -> torch.Tensor:
    """conv1.weight [64][3][7][7] -> wpack bf16 of rpg_stem_conv7x7s2_bn_relu_maxpool_bf16, two operand images back to back
    ([11][2][64][8] for the tile kernel, then [2][3][4][64][8] for the strip-march kernel).  Tile kernel: MFMA step s covers the
    (channel, kernel row) pairs 2s (lanes 0-31) and 2s + 1 (lanes 32-63), 8 kernel columns each (the 8th, and the 22nd pair, are
    zero); lane l of fragment nf holds output channel 32 nf + (l & 31).  Plain bf16 rounding (round-to-nearest-even), no scale
    folded in: like every other bf16 convolution the BatchNorm affine is applied in fp32 to the accumulators."""
    if tuple(w_oihw.shape) != (64, 3, 7, 7):
        raise ValueError("the fused stem kernel is for Conv2d(3, 64, 7)")
    w = w_oihw.float().reshape(64, 21, 7)                                    # [ch][(c, kh)][kw]
    w = torch.nn.functional.pad(w, (0, 1, 0, 1))                             # [ch][22][8]: zero 8th column, zero 22nd row
    w = w.view(2, 32, 11, 2, 8)                                              # [nf][n][s][h][j]
    tile = w.permute(2, 0, 3, 1, 4).reshape(11 * 2 * 64 * 8)                 # [s][nf][l = 32 h + n][j]: the tile kernel (rounds 3-5)
    # second part (round 6, the strip-march kernel): [nf][c][j][l = 32 h + n][t]: kernel rows 2 j + h of channel c (row 7: zeros),
    # taps t = 0: zero, t = 1..7: kernel columns 0..6 -- the window of a pixel starts one input column to the left, at an even one
    v = torch.nn.functional.pad(w_oihw.float(), (1, 0, 0, 1))                # [ch][c][8 kh][8 t]
    v = v.view(2, 32, 3, 4, 2, 8)                                            # [nf][n][c][j][h][t]
    strip = v.permute(0, 2, 3, 4, 1, 5).reshape(2 * 3 * 4 * 64 * 8)          # [nf][c][j][h][n][t]
    return torch.cat([tile, strip]).to(torch.bfloat16).contiguous()


def pack_resnet_bf16(sd: Dict[str, torch.Tensor], prefix: str = "feature_extractor.") -> Tuple[List[torch.Tensor], List[int], List[int]]:
    """bf16 encoder (``rpg_resnet_forward_bf16``): per conv {w_ohwi bf16 (stem Cin padded to 8), scale f32, shift f32},
    then fc weight bf16, fc bias f32.  Weights are rounded to bf16 once (round-to-nearest-even)."""
    t32, blocks, planes = pack_resnet(sd, prefix)
    out: List[torch.Tensor] = []
    for i in range(0, len(t32) - 2, 4):
        w = t32[i]
        if i == 0:                                       # stem: [C][7][7][4] -> [C][7][7][8]
            w = torch.nn.functional.pad(w, (0, 8 - w.shape[-1]))
        out += [w.to(torch.bfloat16).contiguous(), t32[i + 1], t32[i + 2]]
    out += [t32[-2].to(torch.bfloat16).contiguous(), t32[-1]]
    w1 = sd[prefix + "conv1.weight"]
    if tuple(w1.shape) == (64, 3, 7, 7):
        out.append(pack_stem_bf16(w1))                   # optional last tensor: operands of the fused bf16 stem
    return out, blocks, planes


GNN_TENSOR_ORDER = (
    "proj_edge", "edge_mlp.0", "edge_mlp.2", "mlp.0", "mlp.2", "att.gtp", "att.W", "mlp_updating.0", "mlp_updating.2",
    "heads.node", "heads.edge")


def pack_gnn(sd: Dict[str, torch.Tensor], gnn: str = "gnn1.") -> List[torch.Tensor]:
    def lin(name):
        return [sd[name + ".weight"].float().contiguous(), sd[name + ".bias"].float().contiguous()]

    def cat(names):
        return [torch.cat([sd[n + ".weight"].float() for n in names], 0).contiguous(),
                torch.cat([sd[n + ".bias"].float() for n in names], 0).contiguous()]

    t: List[torch.Tensor] = []
    t += lin("proj_edge")
    t += lin(gnn + "edge_model.edge_mlp.0") + lin(gnn + "edge_model.edge_mlp.2")
    t += lin(gnn + "mlp.0") + lin(gnn + "mlp.2")
    t += cat([gnn + "att.g", gnn + "att.theta", gnn + "att.phi"])
    t += lin(gnn + "att.W")
    t += lin(gnn + "mlp_updating.0") + lin(gnn + "mlp_updating.2")
    t += cat(["fc_xyz", "fc_wpqr"])
    t += cat(["fc_xyz_R", "fc_wpqr_R"])
    # column blocks of the concatenated-input Linears as separate row-major matrices (slots 22..25): lets the node terms
    # W_a x[a] be computed once per node and gathered, instead of once per edge
    wp = sd["proj_edge.weight"].float()
    we = sd[gnn + "edge_model.edge_mlp.0.weight"].float()
    wm = sd[gnn + "mlp.0.weight"].float()
    d = wp.shape[1] // 2
    if we.shape[1] == 3 * d and wm.shape[1] == 2 * d and wp.shape[0] == d:
        t.append(torch.cat([wp[:, :d], wp[:, d:]], 0).contiguous())                       # [2D][D]
        t.append(torch.cat([we[:, :d], we[:, d:2 * d], wm[:, :d]], 0).contiguous())       # [3D][D]
        t.append(we[:, 2 * d:].contiguous())                                              # [D][D]
        t.append(wm[:, d:].contiguous())                                                  # [D][D]
    return t


def pack_gnn_bf16(packed: List[torch.Tensor]) -> List[torch.Tensor]:
    """bf16 images of the 10 GEMM weights of the split formulation, in the order rpg_gnn_forward_bf16 expects
    (include/relpose_gnn_hip.h); `packed` is pack_gnn's 26-tensor table."""
    if len(packed) != 26:
        raise ValueError("the bf16 GNN needs the 26-tensor (node/edge split) table")
    order = (22, 23, 24, 4, 25, 8, 10, 12, 14, 16)
    return [packed[i].to(torch.bfloat16).contiguous() for i in order]
