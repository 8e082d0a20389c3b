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
import re
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


class _Unpickler(pickle.Unpickler):
    """PyG classes -> neutral shells; otherwise only tensor-rebuilding helpers of torch / numpy / collections and plain
    builtin containers are resolved (a sample file is data: nothing else has any business being unpickled from it)."""

    def find_class(self, module: str, name: str):
        if module == "torch_geometric" or module.startswith("torch_geometric."):
            return type(name, (_Shell,), {"__module__": module})
        root = module.split(".")[0]
        if root in ("torch", "numpy", "collections") or (module == "builtins" and name in _SAFE_BUILTINS):
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
    """The reference's file list: ``processed/data_*`` under the dataset root (dataset_7Scenes_multi.py:69-75), ordered by
    the integer in the name (equal to lexicographic order for the zero-padded 7-Scenes names)."""
    files = glob.glob(os.path.join(root, "processed", "data_*.pt"))

    def key(p):
        m = re.search(r"data_(\d+)\.pt$", p)
        return int(m.group(1)) if m else -1
    return sorted(files, key=key)


def load_checkpoint_state_dict(path: str, map_location="cpu") -> Dict[str, torch.Tensor]:
    """``torch.load(path)['model_state_dict']`` (test.py:347); a bare state dict is accepted too."""
    ck = torch.load(path, map_location=map_location, weights_only=False)
    if isinstance(ck, dict) and "model_state_dict" in ck:
        return ck["model_state_dict"]
    return ck
