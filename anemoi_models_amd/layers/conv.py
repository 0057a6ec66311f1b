"""Message-passing operators mirroring reference layers/conv.py, backed by the fused HIP edge kernels.

The reference implements these on ``torch_geometric.nn.conv.MessagePassing`` (gather -> message ->
scatter, ~7 materialised ``[E, C]`` temporaries).  Here a conv object is a thin, parameter-free handle
(``GraphTransformerConv``) or the owner of the edge MLP parameters (``GraphConv``); the arithmetic is one
kernel launch over a destination-sorted CSR plan (:class:`anemoi_models_amd.runtime.EdgePlan`).
"""

from __future__ import annotations

from typing import Optional

from torch import Tensor
from torch import nn

from .. import ops
from ..runtime import EdgePlan
from .mlp import MLP


class GraphTransformerConv(nn.Module):
    """Edge attention of the graph transformer (reference layers/conv.py:79-142).

    ``fused`` computes, for every destination ``i`` and head ``h``::

        e_ij = lin_edge(a_ij);  s = q_i . (k_j + e_ij) / sqrt(D);  alpha = segment_softmax_i(s)
        out_i = sum_j alpha (v_j + e_ij)  (+ x_r_i)

    in a single pass (``anemoi_gt_edge_attention``).  ``lin_edge`` is folded into the kernel, so it takes
    the RAW edge attributes in CSR order plus the ``lin_edge`` weight, not a projected ``[E, H, D]`` tensor.
    """

    def __init__(self, out_channels: int, dropout: float = 0.0, **kwargs) -> None:
        super().__init__()
        self.out_channels = out_channels
        self.dropout = dropout

    def fused(self, query: Tensor, key: Tensor, value: Tensor, x_r: Optional[Tensor], edge_attr_csr: Tensor,
              edge_dim: int, w_edge: Tensor, b_edge: Tensor, plan: EdgePlan, num_heads: int) -> Tensor:
        if self.training and self.dropout > 0.0:
            raise NotImplementedError("attention dropout > 0 is not implemented on the MI355X path")
        if query.shape[0] != plan.n_dst or key.shape[0] != plan.n_src:
            raise ValueError(
                f"Encountered tensors with {key.shape[0]} source / {query.shape[0]} destination rows, "
                f"but expected {plan.n_src} / {plan.n_dst}"
            )
        return ops.gt_edge_attention(query, key, value, x_r, edge_attr_csr, edge_dim, w_edge, b_edge, plan.rowptr,
                                     plan.col, num_heads)

    def forward(self, query, key, value, edge_attr, edge_index, size=None):
        raise NotImplementedError(
            "GraphTransformerConv is fused with lin_edge on the MI355X path: call it through "
            "GraphTransformerProcessorBlock / GraphTransformerMapperBlock (or use .fused())"
        )


class GraphConv(nn.Module):
    """Edge-MLP message passing (reference layers/conv.py:27-76): ``e' = MLP(cat[x_i, x_j, e]) + e``, sum over dst."""

    def __init__(self, in_channels: int, out_channels: int, mlp_extra_layers: int = 0, activation: str = "SiLU",
                 **kwargs) -> None:
        super().__init__()
        self.edge_mlp = MLP(3 * in_channels, out_channels, out_channels, n_extra_layers=mlp_extra_layers,
                            activation=activation)
