"""CPU oracle for the relpose-gnn inference hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain PyTorch-CPU (fp32) restatement of the reference forward
``PoseNetX_R2.forward`` in its R3 configuration.  It exists so the parity tests,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` have
something to check the HIP path against on a box where ``/root/reference`` does
not exist.  Nothing in ``relpose-gnn_amd/`` (the product) may import it.

Parity status: PINNED.  The restatement is checked against the reference's own
classes (imported from /root/reference through stand-in torch_geometric /
torch_cluster modules, see ``tests/golden/make_golden.py``) and against the
golden vectors that script wrote into ``tests/golden/``.  The reference holds no
golden vectors or tests of its own for this path.

The oracle is written functionally over a flat ``state_dict`` (same key names
as the reference model) instead of as an nn.Module tree.

Reference lines followed (all under /root/reference/python/niantic/):
  * forward                      modules/posenet.py:1033-1091
  * compute_edge_features        modules/posenet.py:999-1019
  * simpleConvEdge_upt           modules/my_gnn_layer.py:277-311
  * simpleEdgeModel              modules/my_gnn_layer.py:224-239
  * AttentionBlock               modules/att.py:7-34
  * ResNet34 (torchvision 0.9.1 ``models.resnet34``; not vendored in the
    reference, call sites testing/test.py:151 and modules/posenet.py:942-945):
    BasicBlock v1, layers [3,4,6,3], stride on the first conv of a block,
    downsample = conv1x1(stride)+BN, maxpool 3x3/2 pad 1 floor mode, BN eps 1e-5.
  * scatter mean (torch_scatter 2.0.8 ``scatter(reduce='mean')`` reached through
    PyG 2.0.1 ``MessagePassing.propagate``; call site my_gnn_layer.py:301):
    sum over edges with the same target, divided by max(count, 1).
  * FC edge list                 datasets/dataset_7Scenes_multi.py:377-385,418-422
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
RESNET34_BLOCKS = (3, 4, 6, 3)
RESNET34_PLANES = (64, 128, 256, 512)
BN_EPS = 1e-5


# --------------------------------------------------------------------------- #
# graph construction
# --------------------------------------------------------------------------- #
def fc_edge_index(n: int) -> Tensor:
    """Fully-connected directed edge list in the order the reference datasets emit.

    dataset_7Scenes_multi.py:377-385: for offset d = 1..n-1, edges (i -> i+d) for
    i = 0..n-1-d; then (:418-422) the same list with the two rows swapped appended.
    """
    src: List[int] = []
    dst: List[int] = []
    for d in range(1, n):
        for i in range(0, n - d):
            src.append(i)
            dst.append(i + d)
    fwd = torch.tensor([src, dst], dtype=torch.int64)
    return torch.cat([fwd, fwd.flip(0)], dim=1)


def batch_edge_index(n_nodes: int, n_graphs: int) -> Tensor:
    """PyG ``Batch`` collation of ``n_graphs`` identical FC graphs: node ids offset by
    ``n_nodes`` per graph, edge lists concatenated along dim 1 (test.py:193)."""
    ei = fc_edge_index(n_nodes)
    return torch.cat([ei + g * n_nodes for g in range(n_graphs)], dim=1)


# --------------------------------------------------------------------------- #
# ResNet34 encoder
# --------------------------------------------------------------------------- #
def _bn(sd: Dict[str, Tensor], p: str, x: Tensor) -> Tensor:
    # eval-mode batch norm: (x - mean) / sqrt(var + eps) * gamma + beta
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"],
                        sd[p + "weight"], sd[p + "bias"], training=False, eps=BN_EPS)


def _basic_block(sd: Dict[str, Tensor], p: str, x: Tensor, stride: int) -> Tensor:
    out = F.conv2d(x, sd[p + "conv1.weight"], None, stride=stride, padding=1)
    out = F.relu(_bn(sd, p + "bn1.", out))
    out = F.conv2d(out, sd[p + "conv2.weight"], None, stride=1, padding=1)
    out = _bn(sd, p + "bn2.", out)
    if (p + "downsample.0.weight") in sd:
        idt = F.conv2d(x, sd[p + "downsample.0.weight"], None, stride=stride, padding=0)
        idt = _bn(sd, p + "downsample.1.", idt)
    else:
        idt = x
    return F.relu(out + idt)


def resnet34_forward(sd: Dict[str, Tensor], x: Tensor, prefix: str = "feature_extractor.",
                     stages: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """x [N,3,H,W] -> [N,feat_dim].  No ReLU after the final fc (posenet.py:1037)."""
    p = prefix
    x = F.conv2d(x, sd[p + "conv1.weight"], None, stride=2, padding=3)
    x = F.relu(_bn(sd, p + "bn1.", x))
    if stages is not None:
        stages["stem"] = x
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    if stages is not None:
        stages["maxpool"] = x
    for li in range(1, 5):
        bi = 0                       # block count read off the state dict ([3,4,6,3] for ResNet34)
        while f"{p}layer{li}.{bi}.conv1.weight" in sd:
            stride = 2 if (li > 1 and bi == 0) else 1
            x = _basic_block(sd, f"{p}layer{li}.{bi}.", x, stride)
            bi += 1
        if stages is not None:
            stages[f"layer{li}"] = x
    x = x.mean(dim=(2, 3))  # AdaptiveAvgPool2d(1) + flatten
    if stages is not None:
        stages["avgpool"] = x
    x = F.linear(x, sd[p + "fc.weight"], sd[p + "fc.bias"])
    if stages is not None:
        stages["fc"] = x
    return x


# --------------------------------------------------------------------------- #
# GNN pieces
# --------------------------------------------------------------------------- #
def edge_concat(x: Tensor, edge_index: Tensor) -> Tensor:
    """posenet.py:1014-1017: direction-agnostic pair features [x[min(s,t)], x[max(s,t)]]."""
    lo = torch.minimum(edge_index[0], edge_index[1])
    hi = torch.maximum(edge_index[0], edge_index[1])
    return torch.cat([x[lo], x[hi]], dim=1)


def scatter_mean(msg: Tensor, index: Tensor, n: int) -> Tensor:
    """torch_scatter.scatter(msg, index, dim=0, dim_size=n, reduce='mean')."""
    out = torch.zeros(n, msg.shape[1], dtype=msg.dtype)
    out.index_add_(0, index, msg)
    cnt = torch.zeros(n, dtype=msg.dtype)
    cnt.index_add_(0, index, torch.ones(index.shape[0], dtype=msg.dtype))
    return out / cnt.clamp(min=1).unsqueeze(1)


def _mlp2(sd: Dict[str, Tensor], p: str, x: Tensor) -> Tensor:
    # Seq(Linear, ReLU, Linear) with state-dict children "0" and "2"
    h = F.relu(F.linear(x, sd[p + "0.weight"], sd[p + "0.bias"]))
    return F.linear(h, sd[p + "2.weight"], sd[p + "2.bias"])


def attention_block(sd: Dict[str, Tensor], p: str, v: Tensor) -> Tensor:
    """att.py:16-34 applied row-wise: C=in/8 channels, f = phi (outer) theta,
    softmax over the theta axis, y = softmax @ g, z = W y + v."""
    g = F.linear(v, sd[p + "g.weight"], sd[p + "g.bias"])            # [R,C]
    th = F.linear(v, sd[p + "theta.weight"], sd[p + "theta.bias"])    # [R,C]
    ph = F.linear(v, sd[p + "phi.weight"], sd[p + "phi.bias"])        # [R,C]
    f = ph.unsqueeze(2) * th.unsqueeze(1)                              # [R,C,C]  f[r,i,j]=phi_i*theta_j
    a = torch.softmax(f, dim=-1)
    y = torch.bmm(a, g.unsqueeze(2)).squeeze(2)                        # [R,C]
    return F.linear(y, sd[p + "W.weight"], sd[p + "W.bias"]) + v


def gnn_layer(sd: Dict[str, Tensor], p: str, x: Tensor, edge_index: Tensor, e: Tensor,
              stages: Optional[Dict[str, Tensor]] = None, tag: str = "") -> Tuple[Tensor, Tensor]:
    """simpleConvEdge_upt.forward (my_gnn_layer.py:293-311) without PyG."""
    src, dst = edge_index[0], edge_index[1]
    # edge update: edge_mlp(cat[x[src], x[dst], e])             (:296-297, :236-239)
    e_new = _mlp2(sd, p + "edge_model.edge_mlp.", torch.cat([x[src], x[dst], e], dim=1))
    # message: x_j = x[src] (flow source_to_target)              (:304-307)
    msg = _mlp2(sd, p + "mlp.", torch.cat([x[src], e_new], dim=1))
    if stages is not None:
        stages[tag + "edge_update"] = e_new
        stages[tag + "msg_mlp"] = msg
    msg = attention_block(sd, p + "att.", msg)
    # aggregate at the target node, mean                          (:279, :301)
    agg = scatter_mean(msg, dst, x.shape[0])
    # update: mlp_updating(cat[x, agg])                           (:309-311)
    x_new = _mlp2(sd, p + "mlp_updating.", torch.cat([x, agg], dim=1))
    if stages is not None:
        stages[tag + "att"] = msg
        stages[tag + "aggregate"] = agg
        stages[tag + "node_update"] = x_new
    return x_new, e_new


def knn_graph(x: Tensor, k: int, batch: Optional[Tensor] = None) -> Tensor:
    """torch_cluster 1.5.9 ``knn_graph(x, k, batch, loop=False, flow='source_to_target')`` restated (call sites
    posenet.py:1044-1050).  Pinning: torch_cluster itself is not installed and the reference holds no vectors for it, so this
    function cannot be checked against the package; it IS checked against an independent implementation of the same
    published algorithm -- scikit-learn's brute-force ``NearestNeighbors`` per graph (tests/test_oracle_golden.py:
    neighbour sets and nearest-first order on random, ragged (< k+1 nodes), duplicate-row and 2048-wide cases).  The
    algorithm: for every node the k+1 nearest nodes of its own graph by squared Euclidean distance (ties: lower index
    first), the self match removed (``row != col``); row 0 = neighbour (message source), row 1 = the query node (target);
    grouped by target in node order, nearest first."""
    n = x.shape[0]
    batch = torch.zeros(n, dtype=torch.int64) if batch is None else batch
    src: List[int] = []
    dst: List[int] = []
    for i in range(n):
        idx = (batch == batch[i]).nonzero().flatten()
        d = ((x[idx] - x[i]) ** 2).sum(1)
        order = torch.sort(d, stable=True).indices[: k + 1]
        for j in idx[order].tolist():
            if j != i:
                src.append(j)
                dst.append(i)
    return torch.tensor([src, dst], dtype=torch.int64)


def gnn_forward(sd: Dict[str, Tensor], x: Tensor, edge_index: Tensor, gnn_recursion: int = 2,
                stages: Optional[Dict[str, Tensor]] = None, use_AP: bool = True, node_mask: Optional[Tensor] = None,
                edge_mask: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """Everything after the encoder: posenet.py:1052-1091 with use_gnn (edge_index already final).  The reference's
    always-on dropout (posenet.py:1073-1075, F.dropout without training=) is stated with EXPLICIT masks: node_mask [N,D] /
    edge_mask [E,D] hold 0 for a dropped element and 1/(1-p) for a kept one, exactly what F.dropout multiplies by; None =
    droprate 0.  (The random draw itself cannot be pinned across devices; the arithmetic around it can.)"""
    e = F.relu(F.linear(edge_concat(x, edge_index), sd["proj_edge.weight"], sd["proj_edge.bias"]))
    if stages is not None:
        stages["proj_edge"] = e
    for r in range(gnn_recursion):          # same gnn1 weights every recursion (:1061-1069)
        x, e = gnn_layer(sd, "gnn1.", x, edge_index, e, stages, f"r{r}.")
        x, e = F.relu(x), F.relu(e)
    if node_mask is not None:                                          # posenet.py:1073-1075
        x = x * node_mask
    if edge_mask is not None:
        e = e * edge_mask
    h = x if use_AP else edge_concat(x, edge_index)                    # posenet.py:1077-1083
    abs_pose = torch.cat([F.linear(h, sd["fc_xyz.weight"], sd["fc_xyz.bias"]),
                          F.linear(h, sd["fc_wpqr.weight"], sd["fc_wpqr.bias"])], dim=1)
    rel_pose = torch.cat([F.linear(e, sd["fc_xyz_R.weight"], sd["fc_xyz_R.bias"]),
                          F.linear(e, sd["fc_wpqr_R.weight"], sd["fc_wpqr_R.bias"])], dim=1)
    return abs_pose, rel_pose


@torch.no_grad()
def posenet_forward(sd: Dict[str, Tensor], x_flat: Tensor, edge_index: Tensor, img_h: int,
                    gnn_recursion: int = 2, stages: Optional[Dict[str, Tensor]] = None, use_attention: bool = False,
                    use_AP: bool = True, knn: int = -1, k: Optional[int] = None, batch: Optional[Tensor] = None,
                    node_mask: Optional[Tensor] = None, edge_mask: Optional[Tensor] = None
                    ) -> Tuple[Tensor, Tensor, Tensor]:
    """data.x [N,3*H*W], data.edge_index [2,E] -> (abs, rel[E,6], edge_index) -- posenet.py:1033-1091, droprate=0."""
    x = x_flat.view(x_flat.shape[0], 3, img_h, -1).contiguous()       # posenet.py:1035
    feat = resnet34_forward(sd, x, stages=stages)
    if use_attention:                                                  # posenet.py:1040-1041
        feat = attention_block(sd, "att.", feat)
    edge_index_knn = knn_graph(feat, k, batch) if k is not None else None          # :1043-1044
    if knn > 0:                                                        # :1047-1050
        edge_index = knn_graph(feat, knn, batch)
    elif k is not None:
        edge_index = knn_graph(feat, k, batch)
    abs_pose, rel_pose = gnn_forward(sd, feat, edge_index, gnn_recursion, stages, use_AP, node_mask, edge_mask)
    return abs_pose, rel_pose, (edge_index_knn if k is not None else edge_index)   # :1088-1091


# --------------------------------------------------------------------------- #
# caller-side post-processing (test.py:213-251, pose_utils.py:340-348,420-431)
# --------------------------------------------------------------------------- #
def qexp(v):
    import numpy as np
    n = np.linalg.norm(v)
    return np.hstack((np.cos(n), np.sinc(n / np.pi) * v))


def quaternion_angular_error(q1, q2):
    import numpy as np
    d = abs(float(np.dot(q2, q1)))
    d = min(1.0, max(-1.0, d))
    return 2 * np.arccos(d) * 180 / np.pi


def query_pose_from_relative(rel_pose, target, edge_index, ref_node: int = 0):
    """test.py:227-232: take the ``ref_node``-th edge whose target is node 0 and derive the
    query's absolute pose from the neighbour's ground truth and the predicted relative pose."""
    import numpy as np
    edges = np.asarray(edge_index)
    ref = np.argwhere(edges[1] == 0)[ref_node, 0]
    return np.asarray(target)[edges[0, ref]] - np.asarray(rel_pose)[ref]
