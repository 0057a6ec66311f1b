"""Message-passing operators mirroring reference layers/conv.py, backed by the fused HIP edge kernels.

The reference implements these on ``torch_geometric.nn.conv.MessagePassing`` (gather -> message ->
scatter, ~7 materialised ``[E, C]`` temporaries).  Here a conv object is a thin, parameter-free handle
(``GraphTransformerConv``) or the owner of the edge MLP parameters (``GraphConv``); the arithmetic is one
kernel launch over a destination-sorted CSR plan (:class:`anemoi_models_amd.runtime.EdgePlan`).
"""

from __future__ import annotations

from typing import Optional

import torch
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
        self._plans = None  # plan cache of the stand-alone forward (runtime.PlanCache, created on first use)

    def dropout_args(self):
        """``(p, seed, seed_dev)`` of this call: the reference drops attention weights in training mode only
        (layers/conv.py:140).  Seeds as in ``MultiHeadSelfAttention.dropout``: drawn per call from torch's CPU generator
        (``torch.manual_seed`` reproduces the mask); inside a ``runtime.DeviceDropout`` context a per-module constant plus the
        step's device word, so that a captured training step draws a new mask at every replay."""
        import torch

        from .. import runtime

        if not self.training or self.dropout <= 0.0:
            return 0.0, 0, None
        if not 0.0 <= self.dropout <= 1.0:
            raise ValueError(f"dropout probability has to be between 0 and 1, but got {self.dropout}")
        dd = runtime.device_dropout()
        if dd is None:
            return float(self.dropout), int(torch.randint(0, 2**31 - 1, (1,)).item()), None
        if self.__dict__.get("_layer_seed") is None:
            self.__dict__["_layer_seed"] = int(torch.randint(0, 2**31 - 1, (1,)).item())
        return float(self.dropout), self.__dict__["_layer_seed"], dd.word

    def fused(self, query: Tensor, key: Tensor, value: Tensor, x_r: Optional[Tensor], edge_attr_csr: Tensor,
              edge_dim: int, w_edge: Tensor, b_edge: Tensor, plan: EdgePlan, num_heads: int) -> Tensor:
        if self.training and self.dropout > 0.0:  # (no block of the reference constructs its conv with dropout,
            # layers/block.py:339: the folded kernels of the block mirrors carry no mask; ``forward`` below does)
            raise NotImplementedError("dropout > 0 is implemented for GraphTransformerConv.forward (explicit edge "
                                      "features), not for the blocks' folded edge kernels")
        if query.shape[0] != plan.n_dst or key.shape[0] != plan.n_src:
            raise ValueError(
                f"Encountered tensors with {key.shape[0]} source / {query.shape[0]} destination rows, "
                f"but expected {plan.n_src} / {plan.n_dst}"
            )
        try:
            return ops.gt_edge_attention(query, key, value, x_r, edge_attr_csr, edge_dim, w_edge, b_edge, plan.rowptr,
                                         plan.col, num_heads)
        except NotImplementedError:
            if query.shape[1] // num_heads <= 64:
                raise
        # heads beyond the generic kernel's 64 channels that the fast kernels do not take either (bf16 heads of 80 / 96 / 112;
        # heads of 128 with more edge attributes than the fast kernels carry): lin_edge as a GEMM, the conv on explicit edge
        # features with zero-padded heads -- the route the differentiable blocks take for such shapes
        # (autograd.folded_edge_route)
        from .. import autograd

        with torch.no_grad():
            e = autograd.linear(edge_attr_csr[:, :edge_dim].to(query.dtype), w_edge, b_edge)
            return autograd.gt_conv(query, key, value, e, x_r, plan, num_heads)

    def forward(self, query: Tensor, key: Tensor, value: Tensor, edge_attr: Tensor, edge_index: Tensor,
                size=None) -> Tensor:
        """The reference's call (layers/conv.py:98-142): ``query [N_dst, H, D]``, ``key / value [N_src, H, D]``,
        ``edge_attr [E, H, D]`` (= ``lin_edge`` of the raw attributes), ``edge_index [2, E]`` -> ``[N_dst, H, D]``.
        One kernel (``anemoi_gt_conv``) over a destination-sorted plan that is cached per ``edge_index`` tensor; with
        gradients required, the same kernel as an autograd node (``autograd.gt_conv``).  ``dropout > 0`` in training mode
        drops attention weights per (edge, head) as the reference does (layers/conv.py:140), inside the kernels.  The block mirrors do not come
        through here (they fold ``lin_edge`` into the neighbouring GEMMs)."""
        import torch

        from .. import runtime

        if edge_attr is None:
            raise ValueError("GraphTransformerConv needs edge features (the reference adds them to key and value)")
        n_dst, heads, d = query.shape
        n_src = key.shape[0]
        if size is not None and tuple(size) != (n_src, n_dst):
            raise ValueError(f"Encountered tensors with sizes {(n_src, n_dst)}, but expected size {tuple(size)}")
        if self._plans is None:
            self._plans = runtime.PlanCache()
        plan = self._plans.get(edge_index, n_src, n_dst)
        dtype = runtime.compute_dtype(query)
        c = heads * d
        flat = lambda t: (t if t.dtype == dtype else t.to(dtype)).reshape(t.shape[0], c)  # noqa: E731
        kv = torch.cat([flat(key), flat(value)], dim=1)  # one k | v buffer: the kernel gathers both with one row pitch
        p_drop, seed, seed_dev = self.dropout_args()  # (training mode only; the mask is per edge of the SORTED plan and head)
        from .. import autograd  # explicit-edge kernels (anemoi_gt_conv, anemoi_gt_conv_backward_dst / _src) as one autograd node

        if torch.is_grad_enabled() and any(t.requires_grad for t in (query, key, value, edge_attr)):
            edges = autograd.permute_rows(flat(edge_attr).contiguous(), plan.perm.long())
        else:
            edges = flat(edge_attr).index_select(0, plan.perm.long())
        # (bf16 heads of 4 -- config 1's D -- run on the f32 kernels between two casts, head sizes outside the kernels' lane
        #  groups zero-padded: autograd.gt_conv)
        out = autograd.gt_conv(flat(query).contiguous(), kv[:, :c], kv[:, c:], edges, None, plan, heads, p_drop, seed,
                               seed_dev)
        return out.view(n_dst, heads, d).to(query.dtype)


class GraphConv(nn.Module):
    """Edge-MLP message passing (reference layers/conv.py:27-76): ``e' = MLP(cat[x_i, x_j, e]) + e``, sum over dst."""

    def __init__(self, in_channels: int, out_channels: int, mlp_extra_layers: int = 0, activation: str = "SiLU",
                 **kwargs) -> None:
        super().__init__()
        self.edge_mlp = MLP(3 * in_channels, out_channels, out_channels, n_extra_layers=mlp_extra_layers,
                            activation=activation)
        self._plans = None  # plan cache of the stand-alone forward (runtime.PlanCache, created on first use)

    def forward(self, x, edge_attr: Tensor, edge_index: Tensor, size=None):
        """The reference's call (layers/conv.py:62-76): ``x`` is one node tensor or a ``(x_src, x_dst)`` pair, ``edge_attr
        [E, C]`` and ``edge_index [2, E]`` in the caller's edge order -> ``(out [N_dst, C], edges_new [E, C])`` with
        ``edges_new = edge_mlp(cat[x_i, x_j, edge_attr]) + edge_attr`` and ``out`` its sum over the destinations.

        The ``[E, 3C]`` concatenation is never formed: the first Linear splits into two node GEMMs and one edge GEMM,
        the gather kernel adds them and applies the activation, the rest of the MLP runs on the edge rows and a CSR
        segment sum aggregates.  Differentiable (the same autograd nodes the blocks use); the block mirrors call the
        pieces directly so that they can also fuse the node-side GEMMs."""
        import torch

        from .. import autograd, runtime, training

        x_src, x_dst = (x, x) if isinstance(x, Tensor) else x
        n_src, n_dst = x_src.shape[0], x_dst.shape[0]
        if size is not None and tuple(size) != (n_src, n_dst):
            raise ValueError(f"Encountered tensors with sizes {(n_src, n_dst)}, but expected size {tuple(size)}")
        if self._plans is None:
            self._plans = runtime.PlanCache()
        plan = self._plans.get(edge_index, n_src, n_dst)
        dtype = runtime.compute_dtype(x_dst)
        grad = training.wants_grad(self, x_src, x_dst, edge_attr)
        with torch.enable_grad() if grad else torch.no_grad():
            e_csr, inv = training._csr_round_trip(plan, edge_attr, dtype)
            e_new, out = training.gnn_message_pass(self.edge_mlp, training._cast(x_dst, dtype), training._cast(x_src, dtype),
                                                   e_csr, plan)
            return out, autograd.permute_rows(e_new, inv)
