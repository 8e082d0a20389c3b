"""Import shim: the product sources live in ``relpose-gnn_amd/`` (a directory name Python's
``import`` statement cannot spell).  This package forwards its search path there, so
``import relpose_gnn_amd`` and ``from relpose_gnn_amd.posenet import PoseNetX_R2`` load the
modules of ``relpose-gnn_amd/`` under one canonical name."""
import os as _os

_src = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "relpose-gnn_amd")
if not _os.path.isdir(_src):
    raise ImportError(f"relpose_gnn_amd: source directory {_src!r} not found")
__path__ = [_src]
with open(_os.path.join(_src, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_src, "__init__.py"), "exec"), globals())
del _f
