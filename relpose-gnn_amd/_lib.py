"""ctypes binding of the C ABI declared in include/relpose_gnn_hip.h.

The library is the product: there is no CPU or eager-PyTorch fallback.  If the shared object has not been
built (``python relpose-gnn_amd/build.py``) importing a compute entry point raises ``HipLibraryMissing``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

HERE = os.path.dirname(os.path.abspath(__file__))
# RPG_HIP_LIB: another build of the same library (A/B measurements of kernel changes); the product path is the in-tree one
LIB_PATH = os.environ.get("RPG_HIP_LIB") or os.path.join(HERE, "lib", "librelpose_gnn_hip.so")

RPG_OK, RPG_ERR_BAD_ARG, RPG_ERR_LAUNCH, RPG_ERR_WORKSPACE = 0, -1, -2, -3
TIMER_NAMES = ("conv", "linear", "scatter", "attention", "conv_wino", "att_agg")

# every symbol include/relpose_gnn_hip.h declares (tests check the library exports all of them)
SYMBOLS = (
    "rpg_abi_version", "rpg_last_error", "rpg_nchw3_to_nhwc4_f32", "rpg_conv2d_bn_act_nhwc_f32",
    "rpg_maxpool3x3s2_nhwc_f32", "rpg_global_avgpool_nhwc_f32", "rpg_resnet_workspace_bytes",
    "rpg_resnet_forward_f32", "rpg_graph_prepare", "rpg_edge_concat_gather_f32", "rpg_linear_gather_f32",
    "rpg_attention_rows_f32", "rpg_scatter_mean_f32", "rpg_pose_heads_f32", "rpg_gnn_workspace_bytes",
    "rpg_gnn_forward_f32", "rpg_timing_enable", "rpg_timing_read", "rpg_set_tuning", "rpg_knn_graph_f32", "rpg_wino43_transform_weights_f32", "rpg_wino43_weights_floats",
    "rpg_conv3x3_wino43_bn_act_nhwc_f32", "rpg_conv2d_bn_act_nhwc_bf16", "rpg_resnet_bf16_workspace_bytes",
    "rpg_resnet_forward_bf16", "rpg_gnn_forward_bf16", "rpg_f32_to_bf16", "rpg_linear_bf16",
    "rpg_release_scratch", "rpg_timing_read_ex", "rpg_stem_conv7x7s2_bn_relu_maxpool_f32", "rpg_stem_pair_table",
    "rpg_attention_aggregate_f32", "rpg_stem_conv7x7s2_bn_relu_maxpool_bf16", "rpg_stem_conv7x7s2_bn_relu_maxpool_bf16_xbf16",
    "rpg_resnet_forward_bf16_xbf16", "rpg_host_f32_to_bf16", "rpg_basicblock64_bf16", "rpg_linear_gather_ex_f32", "rpg_probe_mfma_bf16", "rpg_host_f32_to_bf16_isa",
)


class HipLibraryMissing(RuntimeError):
    pass


class RpgError(RuntimeError):
    pass


_lib: Optional[C.CDLL] = None
_vp, _i, _sz = C.c_void_p, C.c_int, C.c_size_t


def _declare(lib: C.CDLL) -> None:
    lib.rpg_abi_version.restype = _i
    lib.rpg_last_error.restype = C.c_char_p
    lib.rpg_nchw3_to_nhwc4_f32.argtypes = [_vp, _vp, _i, _i, _i, _vp]
    lib.rpg_conv2d_bn_act_nhwc_f32.argtypes = [_vp] * 6 + [_i] * 10 + [_vp]
    lib.rpg_maxpool3x3s2_nhwc_f32.argtypes = [_vp, _vp, _i, _i, _i, _i, _vp]
    lib.rpg_global_avgpool_nhwc_f32.argtypes = [_vp, _vp, _i, _i, _i, _vp]
    lib.rpg_resnet_workspace_bytes.argtypes = [_i, _i, _i, C.POINTER(_i)]
    lib.rpg_resnet_workspace_bytes.restype = _sz
    lib.rpg_resnet_forward_f32.argtypes = [C.POINTER(_vp), _i, C.POINTER(_i), C.POINTER(_i), _i, _vp, _i, _i, _i, _vp,
                                           _vp, _sz, _vp]
    lib.rpg_graph_prepare.argtypes = [_vp, _vp, C.c_int64, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]
    lib.rpg_edge_concat_gather_f32.argtypes = [_vp, _vp, _i, _i, _vp, _vp]
    lib.rpg_linear_gather_f32.argtypes = [_i, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i), C.POINTER(_i), _vp, _vp,
                                          _vp, _vp, _i, _i, _i, _vp]
    lib.rpg_linear_gather_ex_f32.argtypes = [_i, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i), C.POINTER(_i), C.POINTER(C.c_long),
                                             _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp]
    lib.rpg_attention_rows_f32.argtypes = [_vp, _i, _i, _vp, _vp]
    lib.rpg_scatter_mean_f32.argtypes = [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]
    lib.rpg_attention_aggregate_f32.argtypes = [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]
    lib.rpg_pose_heads_f32.argtypes = [_vp, _vp, _vp, _i, _i, _vp, _vp]
    lib.rpg_gnn_workspace_bytes.argtypes = [_i, _i, _i]
    lib.rpg_gnn_workspace_bytes.restype = _sz
    lib.rpg_gnn_forward_f32.argtypes = [C.POINTER(_vp), _i, _vp, _vp, _vp, C.c_int64, _i, _i, _i, _i, _vp, _vp, _vp, _vp,
                                        _vp, _vp, _sz, _vp]
    lib.rpg_gnn_forward_bf16.argtypes = [C.POINTER(_vp), _i, C.POINTER(_vp), _i, _vp, _vp, _vp, C.c_int64, _i, _i, _i, _i, _vp,
                                         _vp, _vp, _vp, _vp, _vp, _sz, _vp]
    lib.rpg_f32_to_bf16.argtypes = [_vp, _i, _vp, _i, _i, C.c_long, _i, _vp]
    lib.rpg_linear_bf16.argtypes = [_vp] * 7 + [_i, _vp, _i, _i, _i, _i, _vp]
    lib.rpg_timing_enable.argtypes = [_i]
    lib.rpg_set_tuning.argtypes = [_i, _i]
    lib.rpg_conv2d_bn_act_nhwc_bf16.argtypes = [_vp] * 6 + [_i] * 11 + [_vp]
    lib.rpg_basicblock64_bf16.argtypes = [_vp] * 8 + [_i] * 3 + [_vp]
    lib.rpg_resnet_bf16_workspace_bytes.argtypes = [_i, _i, _i, C.POINTER(_i)]
    lib.rpg_resnet_bf16_workspace_bytes.restype = _sz
    lib.rpg_resnet_forward_bf16.argtypes = [C.POINTER(_vp), _i, C.POINTER(_i), C.POINTER(_i), _i, _vp, _i, _i, _i, _vp,
                                            _vp, _sz, _vp]
    lib.rpg_wino43_transform_weights_f32.argtypes = [_vp, _vp, _i, _i, _vp]
    lib.rpg_wino43_weights_floats.argtypes = [_i, _i]
    lib.rpg_wino43_weights_floats.restype = C.c_size_t
    lib.rpg_conv3x3_wino43_bn_act_nhwc_f32.argtypes = [_vp] * 6 + [_i] * 6 + [_vp]
    lib.rpg_knn_graph_f32.argtypes = [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]
    lib.rpg_timing_read.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.POINTER(C.c_double)]
    lib.rpg_timing_read_ex.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.POINTER(C.c_double),
                                       C.POINTER(C.c_double)]
    lib.rpg_release_scratch.argtypes = []
    lib.rpg_stem_conv7x7s2_bn_relu_maxpool_f32.argtypes = [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]
    lib.rpg_stem_pair_table.argtypes = [C.POINTER(_i), C.POINTER(_i)]
    lib.rpg_stem_conv7x7s2_bn_relu_maxpool_bf16.argtypes = [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]
    lib.rpg_stem_conv7x7s2_bn_relu_maxpool_bf16_xbf16.argtypes = [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]
    lib.rpg_resnet_forward_bf16_xbf16.argtypes = lib.rpg_resnet_forward_bf16.argtypes
    lib.rpg_host_f32_to_bf16.argtypes = [_vp, _vp, _sz]
    lib.rpg_host_f32_to_bf16_isa.argtypes = [_vp, _vp, _sz, _i]
    lib.rpg_probe_mfma_bf16.argtypes = [_vp, C.c_long, _i, _vp, _vp]
    for name in SYMBOLS:
        getattr(lib, name)          # AttributeError here = the library does not export a declared symbol


def lib() -> C.CDLL:
    """Load (once) and return the shared library; raises HipLibraryMissing if it has not been built."""
    global _lib
    if _lib is None:
        # RPG_LIB_PATH: a diagnostic build of the same library (ablation / trace variants made by tools/probes/*.sh)
        path = os.environ.get("RPG_LIB_PATH") or LIB_PATH
        return _load(path)
    return _lib


def _load(LIB_PATH: str) -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                f"{LIB_PATH} not found: build it with `python relpose-gnn_amd/build.py` "
                "(the HIP library is the only compute path; there is no fallback)")
        l = C.CDLL(LIB_PATH)
        _declare(l)
        if l.rpg_abi_version() != 1:
            raise HipLibraryMissing(f"{LIB_PATH}: ABI version {l.rpg_abi_version()} != 1, rebuild")
        _lib = l
    return _lib


def check(rc: int, what: str) -> None:
    if rc == RPG_OK:
        return
    if rc == RPG_ERR_BAD_ARG:
        raise ValueError(f"{what}: bad argument (null pointer, non-positive size, unsupported shape or alignment)")
    if rc == RPG_ERR_WORKSPACE:
        raise RpgError(f"{what}: workspace too small")
    msg = lib().rpg_last_error().decode(errors="replace")
    raise RpgError(f"{what}: HIP launch failed: {msg}")


def ptr_array(ptrs: Sequence[Optional[int]]):
    arr = (_vp * len(ptrs))()
    for i, p in enumerate(ptrs):
        arr[i] = p
    return arr


def int_array(vals: Sequence[int]):
    return (_i * len(vals))(*vals)
