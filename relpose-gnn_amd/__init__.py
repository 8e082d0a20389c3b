"""relpose-gnn hot path for AMD MI355X (gfx950): ResNet34 node encoder -> fully-connected
image-graph GNN -> relative-pose heads, as hand-written HIP kernels behind the reference's
``PoseNetX_R2`` nn.Module contract.  See DESIGN.md / INTEGRATION.md at the repo root."""
from .graph import Batch, Data, fc_batch, fc_edge_index  # noqa: F401

__all__ = ["Batch", "Data", "fc_batch", "fc_edge_index"]


def __getattr__(name):          # lazy: importing the package must not require torch.cuda or the built library
    if name in ("PoseNetX_R2",):
        from .posenet import PoseNetX_R2
        return PoseNetX_R2
    if name in ("resnet34", "ResNet"):
        from . import resnet
        return getattr(resnet, name)
    raise AttributeError(name)
