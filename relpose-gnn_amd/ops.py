"""Tensor-level wrappers over the C ABI (one Python function per ``rpg_*`` entry point).

PyTorch is used for device memory and the current stream only: every function takes CUDA (ROCm) fp32
tensors, allocates its output with ``torch.empty`` and launches the HIP kernel on
``torch.cuda.current_stream()``.  CPU tensors are rejected: there is no fallback path.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence, Tuple

import torch

from . import _lib as L


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _req(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not torch.is_tensor(t):
        raise TypeError(f"{name}: expected a tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a tensor on the GPU (the HIP kernels are the only compute path), got {t.device}")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def nchw3_to_nhwc4(x: torch.Tensor) -> torch.Tensor:
    x = _req(x, "x")
    n, c, h, w = x.shape
    if c != 3:
        raise ValueError("expected [N,3,H,W]")
    y = torch.empty((n, h, w, 4), dtype=torch.float32, device=x.device)
    L.check(L.lib().rpg_nchw3_to_nhwc4_f32(_p(x), _p(y), n, h, w, _stream()), "nchw3_to_nhwc4")
    return y


def conv2d_bn_act_nhwc(x: torch.Tensor, w_ohwi: torch.Tensor, scale: Optional[torch.Tensor], shift: Optional[torch.Tensor],
                       residual: Optional[torch.Tensor] = None, stride: int = 1, pad: int = 0, relu: bool = False) -> torch.Tensor:
    x, w_ohwi = _req(x, "x"), _req(w_ohwi, "w_ohwi")
    n, h, w, cin = x.shape
    cout, kh, kw, cin_w = w_ohwi.shape
    if cin_w != cin:
        raise ValueError(f"channel mismatch: x has {cin}, weight has {cin_w}")
    ho, wo = (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1
    y = torch.empty((n, ho, wo, cout), dtype=torch.float32, device=x.device)
    scale = None if scale is None else _req(scale, "scale")
    shift = None if shift is None else _req(shift, "shift")
    residual = None if residual is None else _req(residual, "residual")
    if residual is not None and residual.shape != y.shape:
        raise ValueError("residual shape mismatch")
    L.check(L.lib().rpg_conv2d_bn_act_nhwc_f32(_p(x), _p(w_ohwi), _p(scale), _p(shift), _p(residual), _p(y), n, h, w, cin,
                                                cout, kh, kw, stride, pad, int(relu), _stream()), "conv2d_bn_act_nhwc")
    return y


def wino43_transform_weights(w_ohwi: torch.Tensor) -> torch.Tensor:
    """[Cout][3][3][Cin] -> Winograd F(4,3) weights U [6][Cout][3][Cin] (once per weight load), as one flat buffer of
    rpg_wino43_weights_floats(Cout, Cin) floats (18 Cout Cin; a probe build with the nested 2-D kernel appends its image)."""
    w_ohwi = _req(w_ohwi, "w_ohwi")
    cout, kh, kw, cin = w_ohwi.shape
    if (kh, kw) != (3, 3):
        raise ValueError("Winograd F(4,3) path is for 3x3 kernels")
    floats = int(L.lib().rpg_wino43_weights_floats(cout, cin))
    # [6][Cout][3][Cin] in the product build; a flat buffer when a probe build appends the nested 2-D image behind it
    shape = (6, cout, 3, cin) if floats == 18 * cout * cin else (floats,)
    u = torch.empty(shape, dtype=torch.float32, device=w_ohwi.device)
    L.check(L.lib().rpg_wino43_transform_weights_f32(_p(w_ohwi), _p(u), cout, cin, _stream()), "wino43_transform_weights")
    return u


def conv3x3_wino43_bn_act_nhwc(x: torch.Tensor, u: torch.Tensor, scale: Optional[torch.Tensor], shift: Optional[torch.Tensor],
                               residual: Optional[torch.Tensor] = None, relu: bool = False) -> torch.Tensor:
    x, u = _req(x, "x"), _req(u, "u")
    n, h, w, cin = x.shape
    per = int(L.lib().rpg_wino43_weights_floats(1, 1))            # 18 (42 in a probe build with the nested kernel)
    if u.dim() not in (1, 4) or u.numel() == 0 or u.numel() % (per * cin) or (u.dim() == 4 and tuple(u.shape[::2]) != (6, 3)):
        raise ValueError("u must come from wino43_transform_weights: [6][Cout][3][Cin]")
    if u.dim() == 4:
        if u.shape[3] != cin:
            raise ValueError(f"channel mismatch: x has {cin} channels, u was built for {u.shape[3]}")
        cout = u.shape[1]
    else:                                                          # flat buffer of a probe build: Cout from its size
        cout = u.numel() // (per * cin)
    y = torch.empty((n, h, w, cout), dtype=torch.float32, device=x.device)
    scale = None if scale is None else _req(scale, "scale")
    shift = None if shift is None else _req(shift, "shift")
    residual = None if residual is None else _req(residual, "residual")
    for name, t in (("scale", scale), ("shift", shift)):
        if t is not None and t.numel() != cout:
            raise ValueError(f"{name} must have Cout = {cout} elements, got {t.numel()}")
    if residual is not None and tuple(residual.shape) != (n, h, w, cout):
        raise ValueError("residual shape mismatch")
    L.check(L.lib().rpg_conv3x3_wino43_bn_act_nhwc_f32(_p(x), _p(u), _p(scale), _p(shift), _p(residual), _p(y), n, h, w, cin,
                                                        cout, int(relu), _stream()), "conv3x3_wino43_bn_act_nhwc")
    return y


def conv2d_bn_act_nhwc_bf16(x: torch.Tensor, w_ohwi: torch.Tensor, scale: Optional[torch.Tensor], shift: Optional[torch.Tensor],
                            residual: Optional[torch.Tensor] = None, stride: int = 1, pad: int = 0, relu: bool = False,
                            out_f32: bool = False) -> torch.Tensor:
    """bf16 twin of conv2d_bn_act_nhwc: x / w_ohwi / residual bf16, scale / shift fp32, fp32 accumulation."""
    x, w_ohwi = _req(x, "x", torch.bfloat16), _req(w_ohwi, "w_ohwi", torch.bfloat16)
    n, h, w, cin = x.shape
    cout, kh, kw, cin_w = w_ohwi.shape
    if cin_w != cin:
        raise ValueError(f"channel mismatch: x has {cin}, weight has {cin_w}")
    ho, wo = (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1
    y = torch.empty((n, ho, wo, cout), dtype=torch.float32 if out_f32 else torch.bfloat16, device=x.device)
    scale = None if scale is None else _req(scale, "scale")
    shift = None if shift is None else _req(shift, "shift")
    residual = None if residual is None else _req(residual, "residual", torch.bfloat16)
    L.check(L.lib().rpg_conv2d_bn_act_nhwc_bf16(_p(x), _p(w_ohwi), _p(scale), _p(shift), _p(residual), _p(y), n, h, w, cin,
                                                 cout, kh, kw, stride, pad, int(relu), int(out_f32), _stream()),
            "conv2d_bn_act_nhwc_bf16")
    return y


def basicblock64_bf16(x: torch.Tensor, w1_ohwi: torch.Tensor, scale1: torch.Tensor, shift1: torch.Tensor, w2_ohwi: torch.Tensor,
                      scale2: torch.Tensor, shift2: torch.Tensor) -> torch.Tensor:
    """relu(bn2(conv2(relu(bn1(conv1(x))))) + x) for a 64-channel identity BasicBlock in ONE kernel (the intermediate stays in
    LDS): x bf16 NHWC [n,h,w,64], weights bf16 [64,3,3,64], folded BN scale / shift fp32 [64].  Bit-identical to two
    conv2d_bn_act_nhwc_bf16 calls."""
    x = _req(x, "x", torch.bfloat16)
    w1_ohwi, w2_ohwi = _req(w1_ohwi, "w1_ohwi", torch.bfloat16), _req(w2_ohwi, "w2_ohwi", torch.bfloat16)
    n, h, w, c = x.shape
    if c != 64 or tuple(w1_ohwi.shape) != (64, 3, 3, 64) or tuple(w2_ohwi.shape) != (64, 3, 3, 64):
        raise ValueError("basicblock64_bf16: x [n,h,w,64], weights [64,3,3,64]")
    ps = [_req(t, nm) for t, nm in ((scale1, "scale1"), (shift1, "shift1"), (scale2, "scale2"), (shift2, "shift2"))]
    if any(t.numel() != 64 for t in ps):
        raise ValueError("basicblock64_bf16: scale / shift must have 64 elements")
    y = torch.empty_like(x)
    L.check(L.lib().rpg_basicblock64_bf16(_p(x), _p(w1_ohwi), _p(ps[0]), _p(ps[1]), _p(w2_ohwi), _p(ps[2]), _p(ps[3]), _p(y), n, h, w,
                                          _stream()), "basicblock64_bf16")
    return y


def probe_mfma_bf16(data: str = "relu_like", iters: int = 20000, workgroups: int = 0, device=None) -> float:
    """PFLOP/s that a chip-wide, registers-only stream of v_mfma_f32_32x32x16_bf16 sustains on this device with `data` operands
    ("zeros" | "random" | "relu_like": half zeros, half uniform): the matrix pipe's own ceiling under the power cap
    (rpg_probe_mfma_bf16; measurement aid of bench.py's bf16 roofline).  Timed with events on the current stream."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    g = torch.Generator(device=dev).manual_seed(17)
    n = 65536 * 8
    if data == "zeros":
        v = torch.zeros(n, device=dev)
    elif data == "random":
        v = (torch.rand(n, generator=g, device=dev) - 0.5) * 2e-3
    elif data == "relu_like":
        v = torch.rand(n, generator=g, device=dev) * 2e-3 * (torch.rand(n, generator=g, device=dev) < 0.5)
    else:
        raise ValueError("data: zeros | random | relu_like")
    src = v.bfloat16().contiguous()
    sink = torch.zeros(1, device=dev)
    wgs = workgroups or 2 * torch.cuda.get_device_properties(dev).multi_processor_count
    lib = L.lib()
    L.check(lib.rpg_probe_mfma_bf16(_p(src), max(1, iters // 10), wgs, _p(sink), _stream()), "probe_mfma_bf16")      # warm (clock ramp)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.check(lib.rpg_probe_mfma_bf16(_p(src), iters, wgs, _p(sink), _stream()), "probe_mfma_bf16")
    e1.record()
    e1.synchronize()
    return wgs * 8 * iters * 16 * 32768.0 / (e0.elapsed_time(e1) * 1e-3) / 1e15


def f32_to_bf16(x: torch.Tensor, out: Optional[torch.Tensor] = None, col_off: int = 0) -> torch.Tensor:
    """bf16 image of the fp32 matrix x [rows][cols] (written at column col_off of `out` when given)."""
    x = _req(x, "x")
    rows, cols = x.shape
    if out is None:
        out = torch.empty((rows, cols), dtype=torch.bfloat16, device=x.device)
    out = _req(out, "out", torch.bfloat16)
    L.check(L.lib().rpg_f32_to_bf16(_p(x), cols, _p(out), out.shape[1], col_off, rows, cols, _stream()), "f32_to_bf16")
    return out


def linear_bf16(a: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
                residual: Optional[torch.Tensor] = None, res_idx: Optional[torch.Tensor] = None,
                residual2: Optional[torch.Tensor] = None, res2_idx: Optional[torch.Tensor] = None, relu: bool = False) -> torch.Tensor:
    """out (fp32) = act(a (bf16) @ weight.T (bf16) + bias + residual[res_idx or arange] + residual2[res2_idx]); the
    residual matrices are fp32 with a common row pitch."""
    a, weight = _req(a, "a", torch.bfloat16), _req(weight, "weight", torch.bfloat16)
    m, k = a.shape
    n_out = weight.shape[0]
    if weight.shape[1] != k:
        raise ValueError("a.shape[1] != weight.shape[1]")
    bias = None if bias is None else _req(bias, "bias")
    residual = None if residual is None else _req(residual, "residual")
    residual2 = None if residual2 is None else _req(residual2, "residual2")
    res_idx = None if res_idx is None else _req(res_idx, "res_idx", torch.int64)
    res2_idx = None if res2_idx is None else _req(res2_idx, "res2_idx", torch.int64)
    if residual2 is not None and (residual is None or residual2.shape[1] != residual.shape[1]):
        raise ValueError("residual2 needs residual with the same row pitch")
    ldr = 0 if residual is None else residual.shape[1]
    out = torch.empty((m, n_out), dtype=torch.float32, device=a.device)
    L.check(L.lib().rpg_linear_bf16(_p(a), _p(weight), _p(bias), _p(residual), _p(res_idx), _p(residual2), _p(res2_idx), ldr,
                                    _p(out), m, k, n_out, int(relu), _stream()), "linear_bf16")
    return out


def stem_conv_bn_relu_maxpool(x_nchw: torch.Tensor, wpack: torch.Tensor, shift: torch.Tensor) -> torch.Tensor:
    """conv7x7/2 (3 -> 64) + BN + ReLU + maxpool3x3/2 in one kernel: [N,3,H,W] -> pooled NHWC [N,Hp,Wp,64]; wpack from
    params.pack_stem_pairs (BatchNorm scale folded in), shift [64]."""
    x, wpack, shift = _req(x_nchw, "x_nchw"), _req(wpack, "wpack"), _req(shift, "shift")
    n, c, h, w = x.shape
    if c != 3 or wpack.numel() != (74 + 75) * 2 * 64 or shift.numel() != 64:
        raise ValueError("expected x [N,3,H,W], wpack from params.pack_stem_pairs (19,072 floats), shift [64]")
    hc, wc = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    y = torch.empty((n, (hc - 1) // 2 + 1, (wc - 1) // 2 + 1, 64), dtype=torch.float32, device=x.device)
    L.check(L.lib().rpg_stem_conv7x7s2_bn_relu_maxpool_f32(_p(x), _p(wpack), _p(shift), _p(y), n, h, w, _stream()),
            "stem_conv_bn_relu_maxpool")
    return y


def stem_conv_bn_relu_maxpool_bf16(x_nchw: torch.Tensor, wpack_bf16: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor) -> torch.Tensor:
    """The bf16 encoder's stem in one kernel: fp32 [N,3,H,W] -> bf16 [N,Hp,Wp,64] (wpack_bf16 from params.pack_stem_bf16)."""
    xbf = x_nchw.dtype == torch.bfloat16
    x_nchw, scale, shift = _req(x_nchw, "x_nchw", torch.bfloat16 if xbf else torch.float32), _req(scale, "scale"), _req(shift, "shift")
    wpack_bf16 = _req(wpack_bf16, "wpack_bf16", torch.bfloat16)
    n, c, h, w = x_nchw.shape
    if c != 3 or wpack_bf16.numel() != (11 * 2 + 2 * 3 * 4) * 64 * 8 or scale.numel() != 64 or shift.numel() != 64:
        raise ValueError("expected x [N,3,H,W], wpack_bf16 from params.pack_stem_bf16 (23,552 bf16), scale / shift [64]")
    hc, wc = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    hp, wp = (hc - 1) // 2 + 1, (wc - 1) // 2 + 1
    y = torch.empty((n, hp, wp, 64), dtype=torch.bfloat16, device=x_nchw.device)
    fn = L.lib().rpg_stem_conv7x7s2_bn_relu_maxpool_bf16_xbf16 if xbf else L.lib().rpg_stem_conv7x7s2_bn_relu_maxpool_bf16
    L.check(fn(_p(x_nchw), _p(wpack_bf16), _p(scale), _p(shift), _p(y), n, h, w, _stream()), "stem_conv_bn_relu_maxpool_bf16")
    return y


def maxpool3x3s2_nhwc(x: torch.Tensor) -> torch.Tensor:
    x = _req(x, "x")
    n, h, w, c = x.shape
    y = torch.empty((n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, c), dtype=torch.float32, device=x.device)
    L.check(L.lib().rpg_maxpool3x3s2_nhwc_f32(_p(x), _p(y), n, h, w, c, _stream()), "maxpool3x3s2_nhwc")
    return y


def global_avgpool_nhwc(x: torch.Tensor) -> torch.Tensor:
    x = _req(x, "x")
    n, h, w, c = x.shape
    y = torch.empty((n, c), dtype=torch.float32, device=x.device)
    L.check(L.lib().rpg_global_avgpool_nhwc_f32(_p(x), _p(y), n, h * w, c, _stream()), "global_avgpool_nhwc")
    return y


def graph_prepare(edge_index: torch.Tensor, n: int) -> Dict[str, torch.Tensor]:
    ei = _req(edge_index, "edge_index", torch.int64)
    if ei.dim() != 2 or ei.shape[0] != 2:
        raise ValueError("edge_index must be [2, E]")
    e = ei.shape[1]
    dev = ei.device
    ends = torch.empty((4, e), dtype=torch.int64, device=dev)
    rowptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
    cursor = torch.empty(n, dtype=torch.int32, device=dev)
    perm = torch.empty(e, dtype=torch.int32, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    L.check(L.lib().rpg_graph_prepare(ei.data_ptr(), ei.data_ptr() + 8 * e, 0, e, n, _p(ends), _p(rowptr), _p(cursor), _p(perm),
                                      _p(status), _stream()), "graph_prepare")
    return {"ends": ends, "rowptr": rowptr, "perm": perm, "status": status}


def knn_graph_launch(x: torch.Tensor, k: int, batch: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Enqueue the kNN graph build on the current stream WITHOUT synchronising: returns (edge buffer [2, n*(k+1)] int64,
    meta int32 [2] = (E, graphs-too-large flag), both on the device).  The first E columns of the buffer are the edges once
    the stream has run; knn_graph() below reads E right away, PoseNetX_R2's multi-stream path reads it behind an event."""
    x = _req(x, "x")
    n, d = x.shape
    batch = None if batch is None else _req(batch, "batch", torch.int64)
    cap = n * (k + 1)
    ei = torch.empty((2, cap), dtype=torch.int64, device=x.device)
    cand = torch.empty((n, k + 1), dtype=torch.int32, device=x.device)
    cnt = torch.empty(n, dtype=torch.int32, device=x.device)
    meta = torch.zeros(2, dtype=torch.int32, device=x.device)
    L.check(L.lib().rpg_knn_graph_f32(_p(x), _p(batch), n, d, k, _p(ei), _p(cand), _p(cnt), meta.data_ptr(),
                                      meta.data_ptr() + 4, _stream()), "knn_graph")
    return ei, meta


def knn_graph(x: torch.Tensor, k: int, batch: Optional[torch.Tensor] = None) -> torch.Tensor:
    """torch_cluster.knn_graph(x, k, batch, loop=False): [2, E] int64, row 0 = neighbour, row 1 = query node.
    Synchronises once to learn E (E = n*k unless a graph has fewer than k+1 nodes or duplicate rows)."""
    ei, meta = knn_graph_launch(x, k, batch)
    total, bad = (int(v) for v in meta.tolist())
    if bad:
        raise ValueError("knn_graph: a graph has more than 2048 nodes (unsupported)")
    return ei[:, :total].contiguous()


def edge_concat_gather(x: torch.Tensor, edge_index: torch.Tensor) -> torch.Tensor:
    x, ei = _req(x, "x"), _req(edge_index, "edge_index", torch.int64)
    e, d = ei.shape[1], x.shape[1]
    out = torch.empty((e, 2 * d), dtype=torch.float32, device=x.device)
    L.check(L.lib().rpg_edge_concat_gather_f32(_p(x), _p(ei), e, d, _p(out), _stream()), "edge_concat_gather")
    return out


def linear_gather(sources: Sequence[Tuple[torch.Tensor, Optional[torch.Tensor]]], weight: torch.Tensor,
                  bias: Optional[torch.Tensor], m: int, residual: Optional[torch.Tensor] = None, relu: bool = False) -> torch.Tensor:
    """out[m] = act(cat_k(a_k[idx_k[m]]) @ weight.T + bias (+ residual)); sources = [(a_k, idx_k or None), ...]."""
    ns = len(sources)
    keep = [(_req(a, f"a{i}"), None if ix is None else _req(ix, f"idx{i}", torch.int64)) for i, (a, ix) in enumerate(sources)]
    weight = _req(weight, "weight")
    bias = None if bias is None else _req(bias, "bias")
    residual = None if residual is None else _req(residual, "residual")
    n_out = weight.shape[0]
    if sum(a.shape[1] for a, _ in keep) != weight.shape[1]:
        raise ValueError("sum of source widths != weight.shape[1]")
    out = torch.empty((m, n_out), dtype=torch.float32, device=weight.device)
    a_arr = L.ptr_array([a.data_ptr() for a, _ in keep])
    i_arr = L.ptr_array([None if ix is None else ix.data_ptr() for _, ix in keep])
    ld = L.int_array([a.shape[1] for a, _ in keep])
    wd = L.int_array([a.shape[1] for a, _ in keep])
    L.check(L.lib().rpg_linear_gather_f32(ns, a_arr, i_arr, ld, wd, _p(weight), _p(bias), _p(residual), _p(out), m, n_out,
                                           int(relu), _stream()), "linear_gather")
    return out


def linear_gather_ex(sources: Sequence[Tuple[torch.Tensor, Optional[torch.Tensor]]], weight: torch.Tensor, bias: Optional[torch.Tensor],
                     m: int, residual: Optional[torch.Tensor] = None, res_idx: Optional[torch.Tensor] = None,
                     residual2: Optional[torch.Tensor] = None, res2_idx: Optional[torch.Tensor] = None, relu: bool = False,
                     want_relu_copy: bool = False, widths: Optional[Sequence[int]] = None):
    """rpg_linear_gather_ex_f32: linear_gather with gathered residual rows, an optional max(out, 0) copy and A operands that are
    column blocks of wider tensors (``widths[k]`` < a_k.shape[1]: row pitch a_k.shape[1], the first widths[k] columns are read).
    -> out, or (out, out_relu) with ``want_relu_copy``."""
    ns = len(sources)
    keep = [(_req(a, f"a{i}"), None if ix is None else _req(ix, f"idx{i}", torch.int64)) for i, (a, ix) in enumerate(sources)]
    wd_l = [a.shape[1] for a, _ in keep] if widths is None else [int(v) for v in widths]
    if len(wd_l) != ns or any(w <= 0 or w > a.shape[1] for w, (a, _) in zip(wd_l, keep)):
        raise ValueError("widths: one positive entry per source, at most the source's row pitch")
    weight = _req(weight, "weight")
    bias = None if bias is None else _req(bias, "bias")
    residual = None if residual is None else _req(residual, "residual")
    residual2 = None if residual2 is None else _req(residual2, "residual2")
    res_idx = None if res_idx is None else _req(res_idx, "res_idx", torch.int64)
    res2_idx = None if res2_idx is None else _req(res2_idx, "res2_idx", torch.int64)
    n_out = weight.shape[0]
    if sum(wd_l) != weight.shape[1]:
        raise ValueError("sum of source widths != weight.shape[1]")
    for r_, i_ in ((residual, res_idx), (residual2, res2_idx)):
        if r_ is not None and (r_.shape[1] < n_out or (i_ is None and r_.shape[0] != m) or (i_ is not None and i_.numel() != m)):
            raise ValueError("residual: [rows][>= n_out] with one (gathered) row per output row")
    if residual is not None and residual2 is not None and residual.shape[1] != residual2.shape[1]:
        raise ValueError("residual and residual2 share one row pitch")
    out = torch.empty((m, n_out), dtype=torch.float32, device=weight.device)
    out_relu = torch.empty_like(out) if want_relu_copy else None
    a_arr = L.ptr_array([a.data_ptr() for a, _ in keep])
    i_arr = L.ptr_array([None if ix is None else ix.data_ptr() for _, ix in keep])
    ld = L.int_array([a.shape[1] for a, _ in keep])
    wd = L.int_array(wd_l)
    import ctypes as C
    rows = (C.c_long * ns)(*[a.shape[0] if ix is not None else 0 for a, ix in keep])
    L.check(L.lib().rpg_linear_gather_ex_f32(ns, a_arr, i_arr, ld, wd, rows, _p(weight), _p(bias), _p(residual), _p(res_idx), _p(residual2),
                                              _p(res2_idx), 0 if residual is None else residual.shape[1], _p(out), _p(out_relu), m, n_out,
                                              int(relu), _stream()), "linear_gather_ex")
    return (out, out_relu) if want_relu_copy else out


def attention_rows(gtp: torch.Tensor) -> torch.Tensor:
    gtp = _req(gtp, "gtp")
    r, c3 = gtp.shape
    c = c3 // 3
    y = torch.empty((r, c), dtype=torch.float32, device=gtp.device)
    L.check(L.lib().rpg_attention_rows_f32(_p(gtp), r, c, _p(y), _stream()), "attention_rows")
    return y


def scatter_mean(msg: torch.Tensor, rowptr: torch.Tensor, perm: torch.Tensor, n: int) -> torch.Tensor:
    msg = _req(msg, "msg")
    rowptr, perm = _req(rowptr, "rowptr", torch.int32), _req(perm, "perm", torch.int32)
    e, d = msg.shape
    out = torch.empty((n, d), dtype=torch.float32, device=msg.device)
    L.check(L.lib().rpg_scatter_mean_f32(_p(msg), _p(rowptr), _p(perm), n, e, d, _p(out), _stream()), "scatter_mean")
    return out


def attention_aggregate(gtp: torch.Tensor, msg: torch.Tensor, rowptr: torch.Tensor, perm: torch.Tensor, n: int,
                        bias: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """(ybar [n, c], mbar [n, d]): per-node means of the attention rows of gtp [e, 3c] and of msg [e, d] (+ bias on nodes
    with incoming edges), ascending edge order; see rpg_attention_aggregate_f32."""
    gtp, msg = _req(gtp, "gtp"), _req(msg, "msg")
    rowptr, perm = _req(rowptr, "rowptr", torch.int32), _req(perm, "perm", torch.int32)
    bias = None if bias is None else _req(bias, "bias")
    e, c3 = gtp.shape
    c, d = c3 // 3, msg.shape[1]
    ybar = torch.empty((n, c), dtype=torch.float32, device=gtp.device)
    mbar = torch.empty((n, d), dtype=torch.float32, device=gtp.device)
    L.check(L.lib().rpg_attention_aggregate_f32(_p(gtp), _p(msg), _p(rowptr), _p(perm), _p(bias), n, e, c, d, _p(ybar), _p(mbar),
                                                _stream()), "attention_aggregate")
    return ybar, mbar


def pose_heads(x: torch.Tensor, w6: torch.Tensor, b6: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    x, w6, b6 = _req(x, "x"), _req(w6, "w6"), _req(b6, "b6")
    r, d = x.shape
    if out is None:
        out = torch.empty((r, 6), dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != (r, 6) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != x.device:
        raise ValueError("pose_heads: out must be a contiguous fp32 [rows, 6] tensor on x's device")
    L.check(L.lib().rpg_pose_heads_f32(_p(x), _p(w6), _p(b6), r, d, _p(out), _stream()), "pose_heads")
    return out


def timing_enable(on: bool) -> None:
    L.check(L.lib().rpg_timing_enable(int(on)), "timing_enable")


def timing_read() -> Dict[str, Dict[str, float]]:
    n = len(L.TIMER_NAMES)
    ms, cnt, work, ex = (C.c_double * n)(), (C.c_longlong * n)(), (C.c_double * n)(), (C.c_double * n)()
    L.check(L.lib().rpg_timing_read_ex(ms, cnt, work, ex), "timing_read")
    return {name: {"ms": ms[i], "launches": int(cnt[i]), "work": work[i], "executed": ex[i]}
            for i, name in enumerate(L.TIMER_NAMES)}


def release_scratch() -> None:
    """Free the library-owned split-K scratch pool of the fine-grained entry points (synchronises)."""
    L.check(L.lib().rpg_release_scratch(), "release_scratch")


TUNE_TILE, TUNE_BK, TUNE_EPILOGUE, TUNE_STREAMK, TUNE_WINOGRAD, TUNE_GNN_SPLIT, TUNE_BF16_BK, TUNE_FAST_LOADER, TUNE_WINO_SPLIT, TUNE_BF16_FAST, TUNE_FUSED_STEM, TUNE_WAVES8, TUNE_WINO_SHORT, TUNE_GNN_FUSE_AGG, TUNE_WINO_PERSIST, TUNE_BF16_TILE, TUNE_BF16_DMA, TUNE_BF16_PATCH, TUNE_BF16_WS64, TUNE_SK_MIN_ITS, TUNE_INKERNEL_FIXUP, TUNE_BF16_CHUNK, TUNE_WINO2D, TUNE_BF16_LEAN_EPI, TUNE_BF16_LINEAR_DMA, TUNE_BF16_PERSIST, TUNE_FOLD_K, TUNE_BF16_FUSE_BLOCK, TUNE_BF16_TAIL, TUNE_LIN112, TUNE_FIXUP_PRIO, TUNE_BF16_PAIR = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31


def set_tuning(key: int, value: int) -> None:
    L.check(L.lib().rpg_set_tuning(key, value), "set_tuning")
