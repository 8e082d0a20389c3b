"""Readers for the reference's on-disk formats without PyG installed.

* Graph samples: ``<root>/processed/data_%06d.pt`` (7-Scenes, dataset_7Scenes_multi.py:446) or ``data_%d.pt`` (Cambridge,
  dataset_Cambridge_multi.py:297): ``torch.save`` of a ``torch_geometric.data.Data`` (PyG 2.0.1) holding
  ``x [n, 3*H*W]``, ``edge_index [2, E]``, ``y [n, 6]``, ``edge_attr [E, 6]``.  Unpickling normally needs PyG; here every
  ``torch_geometric.*`` class is mapped to a neutral shell that just keeps its state, and the four tensors are then
  looked up either directly on the object (PyG 1.x layout: attributes in ``__dict__``) or in ``_store._mapping``
  (PyG 2.x layout: a ``GlobalStorage``).
* Checkpoints: ``epoch_%03d.pth.tar`` = ``{'epoch', 'model_state_dict', 'optim_state_dict', 'criterion_state_dict'}``
  (/root/reference/python/niantic/utils/utils.py:22-31; loaded at testing/test.py:347-348).
"""
from __future__ import annotations

import glob
import os
import pickle
from typing import Any, Dict, List

import torch

from .graph import Data

_FIELDS = ("x", "edge_index", "y", "edge_attr")


class _Shell:
    """Stand-in for any torch_geometric class: keeps whatever state the pickle carries."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        elif isinstance(state, tuple) and len(state) == 2 and isinstance(state[1], dict):   # (dict, slots)
            if isinstance(state[0], dict):
                self.__dict__.update(state[0])
            self.__dict__.update(state[1])
        else:
            self.__dict__["_state"] = state


_SAFE_BUILTINS = {"dict", "list", "tuple", "set", "frozenset", "int", "float", "bool", "str", "bytes", "bytearray",
                  "complex", "slice", "range", "object"}
# Everything a tensor-carrying pickle legitimately names, as explicit (module, name) pairs.  A module-prefix rule is not
# enough: `torch.utils.collect_env.run`, `torch.hub.load`, `numpy.testing...runstring` live under the same roots and
# execute code when called from a REDUCE opcode.
_SAFE_GLOBALS = {
    ("collections", "OrderedDict"), ("collections", "defaultdict"),
    ("_codecs", "encode"),                 # how pickle protocol 2 (torch.save's default) writes a `bytes` object: pure data
    ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_parameter"),
    ("torch._utils", "_rebuild_parameter_with_state"), ("torch._tensor", "_rebuild_from_type_v2"),
    ("torch.storage", "UntypedStorage"), ("torch.storage", "TypedStorage"),
    ("torch", "Size"), ("torch", "device"), ("torch", "dtype"), ("torch", "Tensor"), ("torch.nn.parameter", "Parameter"),
    ("torch.serialization", "_get_layout"),
    ("numpy.core.multiarray", "_reconstruct"), ("numpy.core.multiarray", "scalar"),
    ("numpy._core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "scalar"),
    ("numpy", "ndarray"), ("numpy", "dtype"),
}
_TORCH_STORAGES = {f"{t}Storage" for t in ("Float", "Double", "Half", "BFloat16", "Long", "Int", "Short", "Char", "Byte",
                                            "Bool", "ComplexFloat", "ComplexDouble")}
_TORCH_DTYPES = {"float32", "float64", "float16", "bfloat16", "int64", "int32", "int16", "int8", "uint8", "bool",
                 "complex64", "complex128", "float", "double", "half", "long", "int", "short"}


def _load_from_bytes_checked(b):
    """Stand-in for ``torch.storage._load_from_bytes`` (what a storage pickled on its own reduces to).  The original
    calls ``torch.load(io.BytesIO(b), weights_only=False)`` with the DEFAULT pickle module, i.e. a REDUCE of it on attacker
    bytes runs arbitrary code; here the nested stream goes through the same allow-list as the outer one."""
    import io
    return torch.load(io.BytesIO(b), map_location="cpu", pickle_module=_PickleModule, weights_only=False)


class _Unpickler(pickle.Unpickler):
    """PyG classes -> neutral shells; apart from those only an explicit allow-list of tensor-rebuilding helpers, typed
    storages, dtypes and plain containers resolves.  A sample file is data: a pickle that names anything else (any
    callable that could run code from a REDUCE opcode) is refused; ``torch.storage._load_from_bytes`` (which unpickles
    its argument with the unrestricted default module) is replaced by a wrapper that applies this allow-list again."""

    def find_class(self, module: str, name: str):
        if module == "torch_geometric" or module.startswith("torch_geometric."):
            return type(name, (_Shell,), {"__module__": module})
        if (module, name) == ("torch.storage", "_load_from_bytes"):
            return _load_from_bytes_checked
        if ((module, name) in _SAFE_GLOBALS or (module == "builtins" and name in _SAFE_BUILTINS) or
                (module == "torch" and (name in _TORCH_STORAGES or name in _TORCH_DTYPES))):
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"refusing to unpickle {module}.{name} from a graph sample file")


class _PickleModule:
    """What ``torch.load(pickle_module=...)`` needs: Unpickler, load, and the pickle constants."""
    __name__ = "relpose_gnn_amd_pyg_free_pickle"
    Unpickler = _Unpickler
    Pickler = pickle.Pickler
    HIGHEST_PROTOCOL = pickle.HIGHEST_PROTOCOL
    UnpicklingError = pickle.UnpicklingError

    @staticmethod
    def load(f, **kw):
        return _Unpickler(f, **kw).load()


def _lookup(obj: Any, key: str):
    d = getattr(obj, "__dict__", {})
    if key in d and torch.is_tensor(d[key]):
        return d[key]
    store = d.get("_store", None)
    if store is not None:
        m = getattr(store, "__dict__", {}).get("_mapping", None)
        if isinstance(m, dict) and key in m:
            return m[key]
        if key in getattr(store, "__dict__", {}):
            return store.__dict__[key]
    if isinstance(obj, dict) and key in obj:
        return obj[key]
    return None


def load_graph(path: str) -> Data:
    """One pickled PyG ``Data`` sample -> ``relpose_gnn_amd.graph.Data`` (tensors on CPU)."""
    obj = torch.load(path, map_location="cpu", pickle_module=_PickleModule, weights_only=False)
    vals = {k: _lookup(obj, k) for k in _FIELDS}
    if vals["x"] is None or vals["edge_index"] is None:
        raise ValueError(f"{path}: no x / edge_index tensors found (not a relpose-gnn graph sample?)")
    return Data(**vals)


def processed_files(root: str) -> List[str]:
    """The reference's file list: ``processed/data_*`` under the dataset root in the order of ``self.file_list.sort()``
    (dataset_7Scenes_multi.py:69-75,116; dataset_Cambridge_multi.py:72), i.e. LEXICOGRAPHIC on the file name.  For the
    zero-padded 7-Scenes names (``data_%06d.pt``) that is numeric order; for Cambridge's unpadded ``data_%d.pt`` it is
    not (data_10.pt sorts before data_2.pt) and evaluation order / the i-th prediction follow the reference."""
    files = glob.glob(os.path.join(root, "processed", "data_*.pt"))
    return sorted(files, key=os.path.basename)


def load_checkpoint_state_dict(path: str, map_location="cpu") -> Dict[str, torch.Tensor]:
    """``torch.load(path)['model_state_dict']`` (test.py:347); a bare state dict is accepted too.  Loaded through the
    same allow-list as the sample files (a checkpoint is tensors, ints and dicts: utils.py:22-31)."""
    ck = torch.load(path, map_location=map_location, pickle_module=_PickleModule, weights_only=False)
    if isinstance(ck, dict) and "model_state_dict" in ck:
        return ck["model_state_dict"]
    return ck
