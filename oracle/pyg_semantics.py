"""Restatement of the torch-geometric primitives the reference path calls.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

The reference depends on ``torch-geometric>=2.3,<2.5`` (reference
``pyproject.toml:49``), which is not vendored in ``/root/reference`` and not
installed in this image.  The functions below restate the published PyG 2.4
contract of the five utilities the hot path uses; call sites in the reference:

* ``scatter(..., reduce="sum")``        layers/conv.py:74
* ``softmax(src, index, ptr, N)``       layers/conv.py:139
* ``k_hop_subgraph(directed=True)``     distributed/khop_edges.py:43-45
* ``mask_to_index``                     distributed/khop_edges.py:47
* ``bipartite_subgraph``                distributed/khop_edges.py:121-126
"""

from __future__ import annotations

import torch
from torch import Tensor


def scatter_sum(src: Tensor, index: Tensor, dim_size: int) -> Tensor:
    """``torch_geometric.utils.scatter(src, index, dim=0, dim_size, reduce='sum')``.

    PyG: ``src.new_zeros(size).scatter_add_(0, broadcast(index, src), src)``.
    Destinations without any edge stay exactly 0.  On CPU ``scatter_add_`` adds
    in edge order, which defines the summation order of the reference.
    """
    shape = (dim_size,) + tuple(src.shape[1:])
    out = src.new_zeros(shape)
    idx = index.view((-1,) + (1,) * (src.dim() - 1)).expand_as(src)
    return out.scatter_add_(0, idx, src)


def scatter_amax(src: Tensor, index: Tensor, dim_size: int) -> Tensor:
    """``scatter(src, index, 0, dim_size, reduce='max')``; empty groups give 0."""
    shape = (dim_size,) + tuple(src.shape[1:])
    out = src.new_zeros(shape)
    idx = index.view((-1,) + (1,) * (src.dim() - 1)).expand_as(src)
    return out.scatter_reduce_(0, idx, src, reduce="amax", include_self=False)


def segment_softmax(src: Tensor, index: Tensor, num_nodes: int) -> Tensor:
    """``torch_geometric.utils.softmax(src, index, ptr=None, num_nodes=N)``.

    PyG 2.4 (``utils/softmax.py``)::

        src_max = scatter(src.detach(), index, dim, dim_size=N, reduce='max')
        out = (src - src_max.index_select(dim, index)).exp()
        out_sum = scatter(out, index, dim, dim_size=N, reduce='sum') + 1e-16
        return out / out_sum.index_select(dim, index)
    """
    src_max = scatter_amax(src.detach(), index, num_nodes)
    out = (src - src_max.index_select(0, index)).exp()
    out_sum = scatter_sum(out, index, num_nodes) + 1e-16
    return out / out_sum.index_select(0, index)


def k_hop_edge_mask_directed(nodes: Tensor, edge_index: Tensor, num_nodes: int) -> Tensor:
    """Edge mask of ``k_hop_subgraph(nodes, 1, edge_index, directed=True)``.

    With ``flow='source_to_target'`` and one hop the preserved edges are exactly
    those whose target is in ``nodes``; edges are neither relabelled nor
    reordered.
    """
    node_mask = torch.zeros(num_nodes, dtype=torch.bool, device=edge_index.device)
    node_mask[nodes] = True
    return node_mask[edge_index[1]]


def bipartite_dst_mask(dst_nodes: Tensor, edge_index: Tensor, num_dst: int) -> Tensor:
    """Edge mask of ``bipartite_subgraph((all_src, dst_nodes), edge_index, ...)``.

    ``src`` subset is all source nodes, so only the destination test remains;
    ``relabel_nodes`` is False at the reference call site.
    """
    node_mask = torch.zeros(num_dst, dtype=torch.bool, device=edge_index.device)
    node_mask[dst_nodes] = True
    return node_mask[edge_index[1]]
