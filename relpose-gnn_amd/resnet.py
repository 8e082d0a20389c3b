"""torchvision-shaped ResNet (BasicBlock) parameter container whose forward runs the HIP encoder.

torchvision is not a dependency.  The reference builds its encoder with ``torchvision.models.resnet34``
(/root/reference/python/niantic/testing/test.py:151) and hands it to ``PoseNetX_R2`` which replaces ``.avgpool`` and
``.fc`` (/root/reference/python/niantic/modules/posenet.py:942-945).  ``resnet34()`` here returns a module with the same
attribute names and state-dict keys (conv1, bn1, layer{1..4}.{i}.{conv1,bn1,conv2,bn2,downsample.{0,1}}, fc), so
checkpoints load unchanged.  The sub-modules are ordinary ``nn.Conv2d`` / ``nn.BatchNorm2d`` / ``nn.Linear`` objects used
as parameter holders only: the arithmetic is ``rpg_resnet_forward_f32`` (implicit-GEMM MFMA convolutions with the
BatchNorm / residual / ReLU epilogue fused, NHWC activations).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import _lib as L
from .params import pack_resnet, pack_resnet_bf16


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes: int, planes: int, stride: int = 1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
        self.stride = stride

    def forward(self, x):  # pragma: no cover - never used: the whole encoder is one C call
        raise RuntimeError("BasicBlock is a parameter container; call the enclosing ResNet / PoseNetX_R2")


class WorkspacePool:
    """Caller-owned workspaces of the composite C calls, one per concurrent stream slot.  The encoder call and the GNN call
    of a slot run back to back on ONE stream and neither reads what the other left in its workspace (the features travel
    in their own tensor), so both get the SAME buffer, sized for the larger request: the 96-MB split-K / stream-K scratch
    slice that each workspace carries is then held once per slot instead of once per call kind, and the GNN's
    buffers cost nothing on top of the encoder's."""

    def __init__(self):
        self._buf: Dict[object, torch.Tensor] = {}

    def clear(self) -> None:
        self._buf.clear()

    def get(self, slot, nbytes: int, device) -> torch.Tensor:
        # concurrent stream slots get workspaces that start at different offsets within a 2-MiB window: torch hands
        # out 2-MiB-aligned blocks and identically laid-out workspaces would put both streams on the same HBM channels
        skew = ((slot[0] + 3 * slot[1]) % 7 if isinstance(slot, tuple) else int(slot) % 7) * 132 * 1024
        raw = self._buf.get(slot)
        if raw is None or raw.device != device or raw.numel() < nbytes + skew:
            if raw is not None and raw.device == device:
                nbytes = max(nbytes, raw.numel() - skew)      # never shrink: the other call kind of this slot still fits
            raw = torch.empty(nbytes + skew, dtype=torch.uint8, device=device)
            self._buf[slot] = raw
        return raw[skew:]


class EncoderRunner:
    """Packs an encoder state dict for the C ABI and runs it; shared by ``ResNet.forward`` and ``PoseNetX_R2``."""

    def __init__(self, pool: Optional[WorkspacePool] = None):
        self._packed: Optional[Tuple[List[torch.Tensor], List[int], List[int]]] = None
        self._ptrs = None
        self._feat_dim = 0          # rows of fc.weight, read from the state dict itself (not from a position in the packed list)
        self._pool = pool if pool is not None else WorkspacePool()
        self.dtype = "f32"          # "f32": Winograd / direct f32 MFMA kernels; "bf16": bf16 activations + bf16 MFMA

    def invalidate(self) -> None:
        self._packed, self._ptrs = None, None
        self._pool.clear()

    def set_dtype(self, dtype: str) -> None:
        if dtype not in ("f32", "bf16"):
            raise ValueError("encoder dtype must be 'f32' or 'bf16'")
        if dtype != self.dtype:
            self.dtype = dtype
            self.invalidate()

    def ensure_packed(self, state_dict_fn, prefix: str, device) -> None:
        """Pack the weights for the C ABI (layout permutes, BatchNorm folds, Winograd transforms) on the CURRENT stream.
        Callers that spread ``run`` over side streams call this first, on the stream those side streams are forked
        from, so that the packing kernels are ordered before every reader."""
        if self._packed is not None:
            return
        with torch.no_grad():
            sd = {k: v.detach() for k, v in state_dict_fn().items()}
            if any(v.device != device for v in sd.values() if v.is_floating_point()):
                raise RuntimeError("encoder weights and input are on different devices")
            from . import ops
            self._feat_dim = int(sd[prefix + "fc.weight"].shape[0])
            if self.dtype == "bf16":
                self._packed = pack_resnet_bf16(sd, prefix)
            else:
                self._packed = pack_resnet(sd, prefix, wino_fn=ops.wino43_transform_weights)
        self._ptrs = L.ptr_array([None if t is None else t.data_ptr() for t in self._packed[0]])

    def run(self, state_dict_fn, prefix: str, x_nchw: torch.Tensor, slot: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        if not x_nchw.is_cuda:
            raise RuntimeError("the encoder runs on the GPU only (HIP kernels, no CPU fallback); got " + str(x_nchw.device))
        xbf = x_nchw.dtype == torch.bfloat16 and self.dtype == "bf16"     # host-rounded images for the bf16 encoder (evaluate_stream)
        if (x_nchw.dtype != torch.float32 and not xbf) or x_nchw.dim() != 4 or x_nchw.shape[1] != 3:
            raise ValueError("expected fp32 [N,3,H,W] (bf16 is accepted by the bf16 encoder only)")
        lib = L.lib()
        self.ensure_packed(state_dict_fn, prefix, x_nchw.device)
        tensors, blocks, planes = self._packed
        x = x_nchw.contiguous()
        n, _, h, w = x.shape
        feat_dim = self._feat_dim
        planes_c = L.int_array(planes)
        bf16 = self.dtype == "bf16"
        nbytes = (lib.rpg_resnet_bf16_workspace_bytes if bf16 else lib.rpg_resnet_workspace_bytes)(n, h, w, planes_c)
        ws = self._pool.get(slot, nbytes, x.device)      # one workspace per concurrent stream slot (grows, never shrinks)
        feat = out if out is not None else torch.empty((n, feat_dim), dtype=torch.float32, device=x.device)
        if feat.shape != (n, feat_dim) or feat.dtype != torch.float32 or not feat.is_contiguous() or feat.device != x.device:
            raise ValueError("out must be a contiguous fp32 [N, feat_dim] tensor on the input's device")
        fwd = (lib.rpg_resnet_forward_bf16_xbf16 if xbf else lib.rpg_resnet_forward_bf16) if bf16 else lib.rpg_resnet_forward_f32
        rc = fwd(self._ptrs, len(tensors), L.int_array(blocks), planes_c, feat_dim, x.data_ptr(), n, h, w, feat.data_ptr(),
                 ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
        L.check(rc, "resnet_forward")
        return feat


class ResNet(nn.Module):
    def __init__(self, layers: Sequence[int] = (3, 4, 6, 3), planes: Sequence[int] = (64, 128, 256, 512), num_classes: int = 1000):
        super().__init__()
        self.inplanes = planes[0]
        self.conv1 = nn.Conv2d(3, planes[0], 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(planes[0])
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        for li, (c, nb) in enumerate(zip(planes, layers), start=1):
            blocks = []
            for b in range(nb):
                blocks.append(BasicBlock(self.inplanes, c, 2 if (li > 1 and b == 0) else 1))
                self.inplanes = c
            setattr(self, f"layer{li}", nn.Sequential(*blocks))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(planes[-1], num_classes)
        self._runner = EncoderRunner()

    def _apply(self, fn, *a, **k):
        self._runner.invalidate()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._runner.invalidate()
        return super().load_state_dict(*a, **k)

    def refresh_packed(self) -> None:
        """Call after mutating parameters in place (the packed device copies are cached)."""
        self._runner.invalidate()

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [N,3,H,W] -> [N, fc.out_features]; global average pooling regardless of ``self.avgpool``'s output size 1."""
        return self._runner.run(self.state_dict, "", x)


def resnet34(pretrained: bool = False, **kw) -> ResNet:
    """Same call shape as ``torchvision.models.resnet34``.  ``pretrained=True`` cannot download ImageNet weights here
    (no network); load a checkpoint with ``load_state_dict`` instead."""
    if pretrained:
        import warnings
        warnings.warn("resnet34(pretrained=True): no network access, weights stay randomly initialised; "
                      "load a state dict explicitly")
    return ResNet((3, 4, 6, 3), (64, 128, 256, 512), **kw)
