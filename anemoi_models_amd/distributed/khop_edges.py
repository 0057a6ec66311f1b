"""Edge partition by destination ranges (reference distributed/khop_edges.py:24-130), on the tensors' own device.

The reference builds one boolean mask per chunk (``k_hop_subgraph(directed=True)`` / ``bipartite_subgraph``: with one
hop and ``flow="source_to_target"`` both keep exactly the edges whose destination lies in the chunk, un-relabelled, in
original order) and runs ``num_chunks`` passes over the edge list.  Here ONE stable sort of the per-edge chunk id
yields every chunk at once: ``chunk(e) = owner of dst[e]`` under ``tensor_split``'s node ranges, and a stable sort by
that key keeps the original edge order inside a chunk -- the same integer output, bit for bit, for any chunk count.
The destination-sorted CSR of ``runtime.build_edge_plan`` is a refinement of this partition (stable sort by ``dst``
itself): the CSR slots ``rowptr[b_r] .. rowptr[b_{r+1}]`` hold exactly chunk ``r``'s edges, which is what the
node-partitioned forward (``distributed/partition.py``) relies on.
"""

from __future__ import annotations

from typing import List, Optional, Tuple, Union

import torch
from torch import Tensor

from .shapes import split_bounds


def edge_chunk_ids(dst: Tensor, n_dst: int, num_chunks: int) -> Tensor:
    """int64 [E]: index of the ``tensor_split(arange(n_dst), num_chunks)`` range that holds every edge's destination."""
    bounds = torch.tensor(split_bounds(n_dst, num_chunks)[1:], dtype=dst.dtype, device=dst.device)
    return torch.bucketize(dst, bounds, right=True)


def partition_edges_by_dst(dst: Tensor, n_dst: int, num_chunks: int) -> Tuple[Tensor, List[int]]:
    """Stable partition of the edge ids by destination chunk: ``(edge ids, chunk-major; per-chunk counts)``."""
    if dst.numel() > 0 and (int(dst.min()) < 0 or int(dst.max()) >= n_dst):
        raise ValueError(f"edge destinations out of range for {n_dst} destination nodes")
    chunk = edge_chunk_ids(dst, n_dst, num_chunks)
    order = torch.argsort(chunk, stable=True)
    counts = torch.bincount(chunk, minlength=num_chunks).tolist()
    return order, counts


def get_k_hop_edges(nodes: Tensor, edge_attr: Tensor, edge_index: Tensor, num_hops: int = 1) -> Tuple[Tensor, Tensor]:
    """Edges of the directed ``num_hops`` in-neighbourhood of ``nodes`` (reference distributed/khop_edges.py:24-47).

    PyG ``k_hop_subgraph(directed=True)`` contract: hop ``h`` keeps the edges whose target was reached at hop ``h-1``
    (hop 0 = ``nodes``) and continues from their sources; edges are neither relabelled nor reordered.
    Returns ``(edge_attr[mask], edge_index[:, mask])`` like the reference.
    """
    src, dst = edge_index[0], edge_index[1]
    n = int(max(int(edge_index.max()) + 1 if edge_index.numel() else 0, int(nodes.max()) + 1 if nodes.numel() else 0))
    frontier = nodes
    keep = torch.zeros(edge_index.shape[1], dtype=torch.bool, device=edge_index.device)
    for _ in range(num_hops):
        node_mask = torch.zeros(n, dtype=torch.bool, device=edge_index.device)
        node_mask[frontier] = True
        hop = node_mask[dst]
        keep |= hop
        frontier = src[hop]
    return edge_attr[keep], edge_index[:, keep]


def sort_edges_1hop_chunks(num_nodes: Union[int, Tuple[int, int]], edge_attr: Tensor, edge_index: Tensor,
                           num_chunks: int) -> Tuple[List[Tensor], List[Tensor]]:
    """Edge attributes and edge index split into ``num_chunks`` 1-hop neighbourhoods of contiguous destination ranges
    (reference distributed/khop_edges.py:88-130).  ``num_nodes``: int (homogeneous graph) or ``(n_src, n_dst)``."""
    n_dst = num_nodes if isinstance(num_nodes, int) else num_nodes[1]
    order, counts = partition_edges_by_dst(edge_index[1], n_dst, num_chunks)
    edge_index_list = list(edge_index.index_select(1, order).split(counts, dim=1))
    edge_attr_list = list(edge_attr.index_select(0, order).split(counts, dim=0))
    return edge_attr_list, edge_index_list


def sort_edges_1hop_sharding(num_nodes: Union[int, Tuple[int, int]], edge_attr: Tensor, edge_index: Tensor,
                             mgroup: Optional[object] = None) -> Tuple[Tensor, Tensor, list, list]:
    """Edges rearranged rank-major for a model communication group (reference distributed/khop_edges.py:50-85):
    ``(edge_attr, edge_index, per-rank attr shapes, per-rank index shapes)``; identity without a group."""
    if mgroup:
        import torch.distributed as dist

        world = dist.get_world_size(group=mgroup)
        n_dst = num_nodes if isinstance(num_nodes, int) else num_nodes[1]
        order, counts = partition_edges_by_dst(edge_index[1], n_dst, world)
        attr, index = edge_attr.index_select(0, order), edge_index.index_select(1, order)
        attr_shapes = [torch.Size([c, *edge_attr.shape[1:]]) for c in counts]
        index_shapes = [torch.Size([2, c]) for c in counts]
        return attr, index, attr_shapes, index_shapes
    return edge_attr, edge_index, [], []
