"""relpose-gnn hot path for AMD MI355X (gfx950): ResNet34 node encoder -> fully-connected
image-graph GNN -> relative-pose heads, as hand-written HIP kernels behind the reference's
``PoseNetX_R2`` nn.Module contract.  See DESIGN.md / INTEGRATION.md at the repo root."""
from .graph import Batch, Data, fc_batch, fc_edge_index  # noqa: F401

__all__ = ["Batch", "Data", "fc_batch", "fc_edge_index"]
