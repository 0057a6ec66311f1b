"""Per-layer blocks mirroring reference layers/block.py, executed by the gfx950 kernels.

Class names, constructor kwargs, sub-module names (``state_dict`` keys) and forward signatures follow the
reference; the forward bodies are launch sequences over ``libanemoi_amd.so``:

GraphTransformerProcessorBlock (reference layers/block.py:602-635), per call
    LayerNorm -> ONE GEMM for lin_self|lin_query|lin_key|lin_value -> fused edge attention (+ x_r)
    -> projection GEMM (+ x skip in the epilogue) -> LayerNorm -> GEMM+GELU -> GEMM (+ residual).
GraphTransformerMapperBlock (reference layers/block.py:479-550)
    same with separate source / destination node sets (k|v from the sources, self|q from the destinations).
"""

from __future__ import annotations

import os
from abc import ABC
from abc import abstractmethod
from typing import Optional

import torch
from torch import Tensor
from torch import nn

from .. import ops
from .. import runtime
from .. import training
from ..runtime import EdgePlan
from .conv import GraphConv
from .conv import GraphTransformerConv
from .mlp import MLP
from .mlp import NativeSequential
from .mlp import activation_class
from .mlp import linear_native


def inference_num_chunks() -> int:
    """``ANEMOI_INFERENCE_NUM_CHUNKS`` (reference layers/block.py:38-39), read at call time."""
    return int(os.environ.get("ANEMOI_INFERENCE_NUM_CHUNKS", "1"))


def _group_size(group) -> int:
    return 1 if group is None else group.size()


def _as_compute(x: Tensor, dtype: torch.dtype) -> Tensor:
    x = x if x.dtype == dtype else x.to(dtype)
    return x if (x.dim() == 2 and x.stride(1) == 1) else x.contiguous()



def folded_edge_phase(q: Tensor, k: Tensor, v: Tensor, x_r: Optional[Tensor], u: Tensor, edge_attr_csr: Tensor, plan,
                      num_heads: int, up: int, ld_out: Optional[int] = None) -> Tensor:
    """The folded edge phase (``anemoi_gt_edge_attention_folded``: one fused gather -> score -> segment softmax -> weighted
    sum -> ``+ x_r`` pass over the destination-sorted CSR)."""
    runs = edge_runs(plan, q.dtype)
    tiles = None if runs is not None else edge_tiles(plan, q, num_heads, up)
    return ops.gt_edge_attention_folded(q, k, v, x_r, u, edge_attr_csr, plan.rowptr, plan.col, num_heads, up,
                                        ld_out=ld_out, runs=runs, tiles=tiles,
                                        sched=None if runs is not None or tiles is not None else edge_schedule(plan, q))


TILES_DEFAULT = "0"  # ANEMOI_AMD_EDGE_TILES: the LDS-tile edge kernel for mesh graphs (A/B switch; DESIGN 4.2 for the numbers)


def edge_tiles(plan, q: Tensor, num_heads: int, up: int):
    """The tile lists of a plan for the LDS-tile bf16 edge kernel (``EdgePlan.tiles``), else ``None``."""
    if (q.dtype != torch.bfloat16 or not hasattr(plan, "tiles")
            or os.environ.get("ANEMOI_AMD_EDGE_TILES", TILES_DEFAULT) == "0"):
        return None
    return plan.tiles(q.dtype, q.shape[-1], num_heads, up)


def edge_schedule(plan, q: Tensor):
    """The destination schedule of a plan for the scheduled bf16 edge kernel (``EdgePlan.schedule``), else ``None``
    (``ANEMOI_AMD_EDGE_SCHED=0``: the round-robin kernel, A/B)."""
    if q.dtype != torch.bfloat16 or not hasattr(plan, "schedule") or os.environ.get("ANEMOI_AMD_EDGE_SCHED", "1") == "0":
        return None
    return plan.schedule(q.dtype, q.shape[-1])


def set_tile_args(a, tiles, n_src: int, n_edges: int) -> None:
    """The tile fields of an ``anemoi_gt_block_args`` block."""
    a.tile_hdr, a.tile_dst, a.tile_src = tiles.hdr.data_ptr(), tiles.dst.data_ptr(), tiles.src.data_ptr()
    a.tile_slot, a.tile_xcd = tiles.slot.data_ptr(), tiles.xcd.data_ptr()
    a.tile_max_per_xcd, a.tile_src_cap, a.tile_edge_cap = tiles.max_tiles_per_xcd, tiles.src_cap, tiles.edge_cap
    a.n_src, a.n_edges = n_src, n_edges


def edge_runs(plan, dtype):
    """The group (or run) lists of a uniform-degree-3 plan for the bf16 kernels that share source gathers between
    destinations (``EdgePlan.runs3``), else ``None``."""
    if dtype != torch.bfloat16 or not hasattr(plan, "runs3") or os.environ.get("ANEMOI_AMD_EDGE_RUNS", "1") == "0":
        return None
    return plan.runs3()


class EmbeddedRows:
    """Node rows ``h = emb(x)`` handed to a mapper block as the raw features they are embedded from.

    ``x_aug = [x | 1 | 0-pad]`` (the constant-1 column ``one_col`` sits in the K padding and carries the embedding's bias),
    ``emb`` the mapper's ``nn.Linear``.  The block's ``LayerNorm -> Linear`` on these rows then runs as ONE narrow GEMM
    on ``x_aug`` (``runtime.fold_embedded_layer_norm``) instead of a K = C GEMM on ``h``.  ``h`` itself is present only
    when something else needs it (the destination rows: residual of the projection); for source rows that only feed
    k | v it is never formed, and the LayerNorm's row statistics come from a [rows, 256] side product
    (``runtime.embedding_stats_operator``)."""

    def __init__(self, x_aug: Tensor, one_col: int, emb: nn.Linear, h: Optional[Tensor], packed, tag: str) -> None:
        self.x_aug, self.one_col, self.emb, self.h, self._packed, self._tag = x_aug, one_col, emb, h, packed, tag

    @property
    def dtype(self) -> torch.dtype:
        return self.x_aug.dtype

    @property
    def device(self) -> torch.device:
        return self.x_aug.device

    @property
    def shape(self):
        return (self.x_aug.shape[0], self.emb.out_features)

    def stats(self, eps: float) -> Tensor:
        """``{rstd, -mean rstd}`` per row of ``LayerNorm(h)`` (only ``rstd`` is used by the folded product)."""
        if self.h is not None:
            return ops.row_stats(self.h, eps)  # carried by the embedding GEMM's epilogue
        kp = self.x_aug.shape[1]
        t = self._packed.get((self._tag, "embstats", self.dtype, kp, self.one_col), [self.emb.weight, self.emb.bias],
                             lambda: runtime.embedding_stats_operator(self.emb.weight, self.emb.bias, kp, self.one_col,
                                                                      self.dtype))
        return ops.row_stats(ops.linear(self.x_aug, t, None, stats_eps=eps), eps)


class BaseBlock(nn.Module, ABC):
    """Base class for network blocks."""

    @abstractmethod
    def forward(self, x, edge_attr, edge_index, shapes, batch_size, size=None, model_comm_group=None): ...


# =============================================================================================
# Graph transformer blocks
# =============================================================================================
class GraphTransformerBaseBlock(BaseBlock, ABC):
    """Parameters of one graph-transformer layer (reference layers/block.py:289-364)."""

    def __init__(
        self,
        in_channels: int,
        hidden_dim: int,
        out_channels: int,
        edge_dim: int,
        num_heads: int = 16,
        bias: bool = True,
        activation: str = "GELU",
        num_chunks: int = 1,
        update_src_nodes: bool = False,
        **kwargs,
    ) -> None:
        super().__init__(**kwargs)
        self.update_src_nodes = update_src_nodes
        self.out_channels_conv = out_channels // num_heads
        self.num_heads = num_heads
        self.num_chunks = num_chunks
        self.activation = activation
        self.edge_dim = edge_dim
        width = num_heads * self.out_channels_conv

        self.lin_key = nn.Linear(in_channels, width)
        self.lin_query = nn.Linear(in_channels, width)
        self.lin_value = nn.Linear(in_channels, width)
        self.lin_self = nn.Linear(in_channels, width, bias=bias)
        self.lin_edge = nn.Linear(edge_dim, width)
        self.conv = GraphTransformerConv(out_channels=self.out_channels_conv)
        self.projection = nn.Linear(out_channels, out_channels)

        act = activation_class(activation)
        self.node_dst_mlp = nn.Sequential(
            nn.LayerNorm(out_channels), nn.Linear(out_channels, hidden_dim), act(), nn.Linear(hidden_dim, out_channels)
        )
        self.layer_norm1 = nn.LayerNorm(in_channels)
        if self.update_src_nodes:
            self.node_src_mlp = nn.Sequential(
                nn.LayerNorm(out_channels), nn.Linear(out_channels, hidden_dim), act(),
                nn.Linear(hidden_dim, out_channels),
            )
        self._packed = runtime.PackedWeights()
        self._plans = runtime.PlanCache()
        self._dst_mlp: Optional[NativeSequential] = None
        self._src_mlp: Optional[NativeSequential] = None

    # ---- the reference's head / sequence re-sharding helpers (layers/block.py:366-414) -----------------
    def shard_qkve_heads(self, query: Tensor, key: Tensor, value: Tensor, edges: Tensor, shapes: tuple, batch_size: int,
                         model_comm_group=None):
        """``(batch grid) (heads vars) -> (batch grid) heads vars`` for q, k, v and the projected edge features; with a
        model group, the reference's exchange (layers/block.py:366-399): every tensor goes through ``shard_heads`` -- all
        nodes / edges of the group, this rank's heads -- with ``shapes = (source, destination, edge)`` shard shapes.  (The
        model root does not use it: it partitions by mesh node, ``distributed/partition.py``.)"""
        h, d = self.num_heads, self.out_channels_conv
        if _group_size(model_comm_group) <= 1:
            return tuple(t.reshape(t.shape[0], h, d) for t in (query, key, value, edges))
        from ..distributed.transformer import shard_heads

        assert batch_size == 1, "Only batch size of 1 is supported when model is sharded across GPUs"
        shapes_src, shapes_dst, shapes_edges = shapes
        out = []
        for t, shp in ((query, shapes_dst), (key, shapes_src), (value, shapes_src), (edges, shapes_edges)):
            t = t.reshape(1, t.shape[0], h, d).permute(0, 2, 1, 3)  # batch heads grid vars
            t = shard_heads(t, shapes=shp, mgroup=model_comm_group)
            out.append(t[0].permute(1, 0, 2).contiguous())  # grid heads vars
        return tuple(out)

    def shard_output_seq(self, out: Tensor, shapes: tuple, batch_size: int, model_comm_group=None) -> Tensor:
        """``(batch grid) heads vars -> (batch grid) (heads vars)`` (reference layers/block.py:401-414); with a model
        group the destination rows come back through ``shard_sequence`` (all heads, this rank's rows)."""
        if _group_size(model_comm_group) <= 1:
            return out.reshape(out.shape[0], -1)
        from ..distributed.transformer import shard_sequence

        t = out.permute(1, 0, 2).unsqueeze(0)  # batch heads grid vars
        t = shard_sequence(t, shapes=shapes[1], mgroup=model_comm_group)
        return t[0].permute(1, 0, 2).reshape(t.shape[2], -1).contiguous()

    # ---- packed parameters -------------------------------------------------------------------
    def _cat_linear(self, tag: str, layers, dtype):
        w = self._packed.get((tag, "w", dtype), [l.weight for l in layers],
                             lambda: runtime.pack_weight([l.weight for l in layers], dtype))
        b = self._packed.get((tag, "b"), [l.bias for l in layers],
                             lambda: runtime.pack_bias([l.bias for l in layers], [l.out_features for l in layers],
                                                       layers[0].weight.device))
        return w, b

    # ---- LayerNorm -> Linear pairs: row statistics + anemoi_linear_ln (the normalised activation is never stored)
    @staticmethod
    def _ln_begin(ln: nn.LayerNorm, x: Tensor):
        """Handle of ``LayerNorm(x)`` for ``_ln_linear``: ``(x, stats, ln)`` when folding, else ``(LN(x), None, None)``."""
        if isinstance(x, EmbeddedRows):
            return x, x.stats(ln.eps), ln
        if runtime.ln_fold_enabled(x.dtype):
            return x, ops.row_stats(x, ln.eps), ln
        return ops.layer_norm(x, runtime.f32c(ln.weight), runtime.f32c(ln.bias), ln.eps), None, None

    def _ln_linear(self, handle, tag: str, plain, rows, params, **kw) -> Tensor:
        """``Linear(LayerNorm(x))``.  ``plain()`` -> packed ``(w, b)`` of the unfolded route; ``rows()`` -> f32
        ``(weight rows [N, K], bias [N] or None)`` that the fold scales by the LayerNorm weight (cached per dtype)."""
        xin, stats, ln = handle
        if isinstance(xin, EmbeddedRows):
            # embedding -> LayerNorm -> Linear as one product on the raw features (K = padded feature count, not C)
            xa = xin.x_aug
            wf, bf, zero = self._packed.get(
                (tag, "embfold", xa.dtype, xa.shape[1], xin.one_col, xin._tag),
                list(params) + [ln.weight, ln.bias, xin.emb.weight, xin.emb.bias],
                lambda: runtime.fold_embedded_layer_norm(*rows(), ln.weight, ln.bias, xin.emb.weight, xin.emb.bias,
                                                         xa.shape[1], xin.one_col, xa.dtype))
            return ops.linear(xa, wf, bf, ln=(stats, zero), **kw)
        if stats is None:
            w, b = plain()
            return ops.linear(xin, w, b, **kw)
        wf, bf, cs = self._packed.get(
            (tag, "lnfold", xin.dtype), list(params) + [ln.weight, ln.bias],
            lambda: runtime.fold_layer_norm(*rows(), ln.weight, ln.bias, xin.dtype))
        return ops.linear(xin, wf, bf, ln=(stats, cs), **kw)

    def _cat_rows(self, layers):
        """f32 ``(cat of weights, cat of biases)`` of a list of Linear layers (input of the LayerNorm fold)."""
        w = torch.cat([l.weight.detach().float() for l in layers], dim=0)
        b = runtime.pack_bias([l.bias for l in layers], [l.out_features for l in layers], w.device)
        return w, b

    def _folded_rows(self, lead_layers, up: int):
        """As ``_cat_rows`` with the ``W_u`` rows of the lin_edge fold appended (q/k/v side of the lin_edge fold, ``_folded_in``)."""
        w, b = self._cat_rows(lead_layers)
        wu, bu = self._query_fold(up)
        return torch.cat([w, wu], dim=0), torch.cat([b, bu], dim=0)

    def _edge_params(self):
        return runtime.f32c(self.lin_edge.weight), runtime.f32c(self.lin_edge.bias)

    # ---- lin_edge folded into the neighbouring GEMMs (see include/anemoi_amd.h: anemoi_gt_edge_attention_folded)
    def fold_width(self, dtype) -> Optional[int]:
        """Per-head width ``up`` of the folded edge representation (``edge_dim`` attributes + the constant 1,
        rounded up to 4), or ``None`` when this shape has to use the unfolded kernel."""
        if os.environ.get("ANEMOI_AMD_EDGE_FOLD", "1") == "0":
            return None
        vec = 16 // torch.empty((), dtype=dtype).element_size()
        d, h = self.out_channels_conv, self.num_heads
        if d % vec != 0 or (d // vec) not in (1, 2, 4, 8, 16):
            return None
        up = ops.round_up(self.edge_dim + 1, 4)
        if up > 16 or (h * up) % vec != 0 or self.lin_self.in_features % vec != 0:
            return None
        return up

    def edge_layout(self, dtype):
        """(row width, index of the constant-1 column or -1) of the CSR edge-attribute matrix this block consumes."""
        up = self.fold_width(dtype)
        return (ops.round_up(self.edge_dim, 4), -1) if up is None else (up, self.edge_dim)

    def _edge_fold(self, up: int) -> Tensor:
        """``W_e' = [W_e | b_e | 0]`` as ``[H, D, up]`` in f32."""
        h, d = self.num_heads, self.out_channels_conv
        we = torch.zeros((h * d, up), dtype=torch.float32, device=self.lin_edge.weight.device)
        we[:, : self.edge_dim] = self.lin_edge.weight.detach().float()
        we[:, self.edge_dim] = self.lin_edge.bias.detach().float()
        return we.view(h, d, up)

    def _query_fold(self, up: int):
        """Rows / bias that make the q GEMM also emit ``u[n, h, a] = sum_{c in h} W_e'[c, a] q[n, c]``."""
        h, d = self.num_heads, self.out_channels_conv
        weh = self._edge_fold(up)
        wq = self.lin_query.weight.detach().float().view(h, d, -1)
        wu = torch.einsum("hda,hdc->hac", weh, wq).reshape(h * up, -1)
        bu = torch.einsum("hda,hd->ha", weh, self.lin_query.bias.detach().float().view(h, d)).reshape(h * up)
        return wu, bu

    def _projection_fold(self, up: int) -> Tensor:
        """Columns that make ``projection`` absorb ``W_e' t``: ``W_t[o, (h, a)] = sum_{c in h} W_p[o, c] W_e'[c, a]``."""
        h, d = self.num_heads, self.out_channels_conv
        wp = self.projection.weight.detach().float()
        return torch.einsum("ohd,hda->oha", wp.view(wp.shape[0], h, d), self._edge_fold(up)).reshape(wp.shape[0],
                                                                                                       h * up)

    def _folded_in(self, tag: str, lead_layers, dtype, up: int):
        """Packed ``[lead_layers..., W_u]`` weight + bias (q/k/v side of the lin_edge fold)."""
        edge_q = [self.lin_edge.weight, self.lin_edge.bias, self.lin_query.weight, self.lin_query.bias]
        w_in = self._packed.get((tag, "w", dtype, up), [l.weight for l in lead_layers] + edge_q,
                                lambda: runtime.pack_weight([l.weight for l in lead_layers] + [self._query_fold(up)[0]],
                                                            dtype))
        b_in = self._packed.get((tag, "b", up), [l.bias for l in lead_layers] + edge_q,
                                lambda: torch.cat([runtime.pack_bias([l.bias for l in lead_layers],
                                                                     [l.out_features for l in lead_layers],
                                                                     self.lin_edge.weight.device),
                                                   self._query_fold(up)[1]]).contiguous())
        return w_in, b_in

    def _folded_out(self, dtype, up: int):
        """Packed ``[W_p | W_t]`` weight + bias (projection side of the lin_edge fold)."""
        edge_p = [self.lin_edge.weight, self.lin_edge.bias, self.projection.weight]
        w_out = self._packed.get(("projf", "w", dtype, up), edge_p,
                                 lambda: runtime.pack_weight_cols([self.projection.weight.detach().float(),
                                                                   self._projection_fold(up)], dtype))
        return w_out, runtime.f32c(self.projection.bias)

    def _fold_params(self, lead_layers):
        return ([l.weight for l in lead_layers] + [l.bias for l in lead_layers]
                + [self.lin_edge.weight, self.lin_edge.bias, self.lin_query.weight, self.lin_query.bias])

    def _mlp_ln_eps(self, which: str, dtype) -> Optional[float]:
        """Epsilon of the LayerNorm that opens the node MLP when its statistics can ride on the producing GEMM."""
        mlp = self.node_dst_mlp if which == "dst" else self.node_src_mlp
        first = mlp[0] if isinstance(mlp, nn.Sequential) else None
        return first.eps if isinstance(first, nn.LayerNorm) and runtime.ln_fold_enabled(dtype) else None

    def _next_ln_eps(self, dtype) -> Optional[float]:
        """Processor blocks are stacked: the output enters the next block's ``layer_norm1`` (same construction)."""
        ln1 = getattr(self, "layer_norm1", None)
        return ln1.eps if isinstance(ln1, nn.LayerNorm) and runtime.ln_fold_enabled(dtype) else None

    def block_abi_operands(self, dtype: torch.dtype, lead_layers, tag: str):
        """Packed operands of this block for the block-level C entry points (``anemoi_gt_block_tail`` /
        ``anemoi_gt_processor_block_forward``) -- the very tensors the op-by-op route caches and passes --, or ``None`` when
        the block cannot take that route (f32, LayerNorm fold off, unfolded edge kernel, another node MLP shape).
        ``lead_layers`` / ``tag``: the Linear layers whose rows open the LayerNorm-folded input product (processor block:
        lin_self, lin_query, lin_key, lin_value under "sqkvu")."""
        up = self.fold_width(dtype)
        if up is None or not runtime.ln_fold_enabled(dtype):
            return None
        if self._dst_mlp is None:
            self._dst_mlp = NativeSequential(self.node_dst_mlp)
        mlp = self._dst_mlp.folded_pair(dtype)
        if mlp is None:
            return None
        ln = self.layer_norm1
        w_in, b_in, cs_in = self._packed.get(
            (tag, "lnfold", dtype), list(self._fold_params(lead_layers)) + [ln.weight, ln.bias],
            lambda: runtime.fold_layer_norm(*self._folded_rows(lead_layers, up), ln.weight, ln.bias, dtype))
        w_proj, b_proj = self._folded_out(dtype, up)
        eps_mlp, act, w1, b1, cs1, w2, b2 = mlp
        return dict(up=up, w_in=w_in, b_in=b_in, cs_in=cs_in, w_proj=w_proj, b_proj=b_proj, eps_mlp=eps_mlp, act=act, w_fc1=w1,
                    b_fc1=b1, cs_fc1=cs1, w_fc2=w2, b_fc2=b2, eps_ln1=ln.eps)

    def _block_tail(self, q: Tensor, k: Tensor, v: Tensor, x_r: Tensor, u: Tensor, edge_attr_csr: Tensor, plan, res: Tensor,
                    up: int, out_stats_eps: Optional[float]) -> Optional[Tensor]:
        """Edge phase -> projection (+ ``res``) -> node MLP (+ skip) as ONE call of ``anemoi_gt_block_tail`` (the launches
        and packed weights of ``folded_edge_phase`` + ``ops.linear`` + ``_node_mlp``; bit-identical), or ``None`` when this
        block / call has to go op by op (f32, LayerNorm fold off, another MLP shape, bench.py's per-kernel timing pass)."""
        import ctypes

        from .. import _lib

        dtype = q.dtype
        if (ops.PROFILE is not None or dtype != torch.bfloat16 or not q.is_cuda or plan.num_edges == 0 or q.shape[0] == 0
                or not self.block_abi or self._mlp_ln_eps("dst", dtype) is None):
            return None
        if self._dst_mlp is None:
            self._dst_mlp = NativeSequential(self.node_dst_mlp)
        mlp = self._dst_mlp.folded_pair(dtype)
        if mlp is None or not (res.dim() == 2 and res.stride(1) == 1):
            return None
        eps_mlp, act, w1, b1, cs1, w2, b2 = mlp
        wp, bp = self._folded_out(dtype, up)
        n, c, h = q.shape[0], q.shape[1], self.num_heads
        dev = q.device
        att = torch.empty((n, wp.shape[1]), dtype=dtype, device=dev)
        if wp.shape[1] > c + h * up:
            att[:, c + h * up:].zero_()
        y, hid, out = (torch.empty((n, w), dtype=dtype, device=dev) for w in (c, w1.shape[0], c))
        stats = torch.empty((2 if out_stats_eps is not None else 1, n, 2), dtype=torch.float32, device=dev)
        ws = torch.empty((n * max(c // 128, 1), 2), dtype=torch.float32, device=dev)
        a = _lib.GtBlockArgs()
        a.struct_bytes, a.n_dst, a.dtype = ctypes.sizeof(_lib.GtBlockArgs), n, ops.dtype_code(dtype)
        a.C, a.H, a.up, a.hidden, a.act, a.k_proj = c, h, up, w1.shape[0], _lib.ACT_CODES[act], wp.shape[1]
        a.eps_mlp, a.eps_out = eps_mlp, (out_stats_eps if out_stats_eps is not None else 0.0)
        a.q, a.k, a.v, a.x_r, a.u = q.data_ptr(), k.data_ptr(), v.data_ptr(), x_r.data_ptr(), u.data_ptr()
        a.ldq, a.ldkv, a.ldr, a.ldu = ops._ld(q), ops._ld(k), ops._ld(x_r), ops._ld(u)
        a.edge_attr, a.rowptr, a.col = edge_attr_csr.data_ptr(), plan.rowptr.data_ptr(), plan.col.data_ptr()
        runs = edge_runs(plan, dtype)
        if runs is not None:
            a.run_ptr, a.run_perm, a.n_runs = runs[0].data_ptr(), runs[1].data_ptr(), runs[0].shape[0] - 1
            if len(runs) == 3:  # groups: the destination list, and the row count of k / v for the 4-GiB check
                a.run_dst, a.n_src = runs[2].data_ptr(), k.shape[0]
        else:  # (beyond 32-bit attribute-row offsets the entry point takes the plain kernel by itself: n_edges states the size)
            tiles = edge_tiles(plan, q, h, up)
            if tiles is not None:
                set_tile_args(a, tiles, k.shape[0], edge_attr_csr.shape[0])
            sched = None if tiles is not None else edge_schedule(plan, q)
            if sched is not None:
                a.sched, a.sched_slots, a.sched_steps, a.n_src = sched.data_ptr(), sched.shape[1], sched.shape[2], k.shape[0]
                a.n_edges = edge_attr_csr.shape[0]
        a.att, a.ld_att = att.data_ptr(), wp.shape[1]
        a.w_proj, a.b_proj = wp.data_ptr(), ops._ptr(bp)
        a.res, a.ld_res, a.y, a.y_stats = res.data_ptr(), ops._ld(res), y.data_ptr(), stats[0].data_ptr()
        a.w_fc1, a.b_fc1, a.cs_fc1, a.h = w1.data_ptr(), ops._ptr(b1), cs1.data_ptr(), hid.data_ptr()
        a.w_fc2, a.b_fc2, a.out = w2.data_ptr(), ops._ptr(b2), out.data_ptr()
        a.out_stats = stats[1].data_ptr() if out_stats_eps is not None else None
        a.stats_ws, a.stats_ws_bytes = ws.data_ptr(), ws.numel() * 4
        _lib.check(_lib.load().anemoi_gt_block_tail(ctypes.byref(a), ops._stream()), "anemoi_gt_block_tail")
        if out_stats_eps is not None:
            ops._carry_stats(out, out_stats_eps, stats[1])
        return out

    block_abi = True  # False: op by op (tests compare the two routes)

    def _node_mlp(self, y: Tensor, which: str, num_chunks: int, out_stats_eps: Optional[float] = None) -> Tensor:
        """``mlp(y) + y`` with mlp = LayerNorm, Linear, act, Linear; optionally in row chunks (bounded hidden buffer)."""
        if which == "dst":
            if self._dst_mlp is None:
                self._dst_mlp = NativeSequential(self.node_dst_mlp)
            run = self._dst_mlp
        else:
            if self._src_mlp is None:
                self._src_mlp = NativeSequential(self.node_src_mlp)
            run = self._src_mlp
        if num_chunks <= 1:
            return run(y, residual=y, out_stats_eps=out_stats_eps)
        return torch.cat([run(c, residual=c) for c in y.tensor_split(num_chunks, dim=0) if c.shape[0] > 0], dim=0)

    def _check_channels(self, dtype) -> None:
        mult = ops.k_multiple(dtype)
        width = self.num_heads * self.out_channels_conv
        if width % mult != 0:
            raise NotImplementedError(
                f"hidden width {width} must be a multiple of {mult} for {dtype} on the MI355X path"
            )

    def _edge_inputs(self, edge_attr: Tensor, edge_index: Tensor, n_src: int, n_dst: int, dtype):
        if edge_attr.shape[1] != self.edge_dim:
            raise ValueError(f"edge_attr has {edge_attr.shape[1]} features, lin_edge expects {self.edge_dim}")
        if edge_attr.shape[0] != edge_index.shape[1]:
            raise ValueError(f"edge_attr has {edge_attr.shape[0]} rows for {edge_index.shape[1]} edges")
        plan = self._plans.get(edge_index, n_src, n_dst)
        return plan, ops.edge_attr_csr(edge_attr, None, plan.perm, *self.edge_layout(dtype))

    @abstractmethod
    def forward(self, x, edge_attr, edge_index, shapes, batch_size, model_comm_group=None, size=None): ...


class GraphTransformerProcessorBlock(GraphTransformerBaseBlock):
    """Graph transformer layer on one node set (reference layers/block.py:553-635)."""

    def native(self, x: Tensor, edge_attr_csr: Tensor, plan: EdgePlan, halo=None) -> Tensor:
        """x ``[N, C]`` in the compute dtype, edge attributes already in CSR order.  Returns the new nodes.

        With ``halo`` (node-partitioned run) ``x`` holds this rank's rows only; the k|v rows of the halo sources are
        fetched from their owners by one all-to-all-v and appended behind the own rows (the plan's source index space).
        """
        dtype = x.dtype
        self._check_channels(dtype)
        c = self.num_heads * self.out_channels_conv
        xh = self._ln_begin(self.layer_norm1, x)
        up = self.fold_width(dtype)
        if halo is not None:
            if up is None:
                raise NotImplementedError("node-partitioned blocks need the folded edge kernel")
            n_own = x.shape[0]
            kv_layers, sq_layers = [self.lin_key, self.lin_value], [self.lin_self, self.lin_query]
            kv = torch.empty((n_own + halo.n_recv, 2 * c), dtype=dtype, device=x.device)
            self._ln_linear(xh, "kv", lambda: self._cat_linear("kv", kv_layers, dtype),
                            lambda: self._cat_rows(kv_layers), [l.weight for l in kv_layers] + [l.bias for l in kv_layers],
                            out=kv[:n_own])
            pending = halo.start(kv, n_own)  # xGMI transfer of the halo k|v rows ...
            wpf, bp = self._folded_out(dtype, up)
            sq = self._ln_linear(xh, "squ", lambda: self._folded_in("squ", sq_layers, dtype, up),
                                 lambda: self._folded_rows(sq_layers, up),
                                 self._fold_params(sq_layers))  # ... overlapped with the x_r | q | u GEMM
            halo.finish(pending)
            done = self._block_tail(sq[:, c:2 * c], kv[:, :c], kv[:, c:], sq[:, :c], sq[:, 2 * c:], edge_attr_csr, plan, x, up,
                                    self._next_ln_eps(dtype))
            if done is not None:
                return done
            att = folded_edge_phase(sq[:, c:2 * c], kv[:, :c], kv[:, c:], sq[:, :c], sq[:, 2 * c:], edge_attr_csr, plan,
                                    self.num_heads, up, ld_out=wpf.shape[1])
            y = ops.linear(att, wpf, bp, residual=x, stats_eps=self._mlp_ln_eps("dst", dtype))
            return self._node_mlp(y, "dst", 1, out_stats_eps=self._next_ln_eps(dtype))
        all4 = [self.lin_self, self.lin_query, self.lin_key, self.lin_value]
        if up is not None:
            wpf, bp = self._folded_out(dtype, up)
            sq = self._ln_linear(xh, "sqkvu", lambda: self._folded_in("sqkvu", all4, dtype, up),
                                 lambda: self._folded_rows(all4, up),
                                 self._fold_params(all4))  # [N, 4C + H*up] = x_r | q | k | v | u
            done = self._block_tail(sq[:, c:2 * c], sq[:, 2 * c:3 * c], sq[:, 3 * c:4 * c], sq[:, :c], sq[:, 4 * c:],
                                    edge_attr_csr, plan, x, up, self._next_ln_eps(dtype))
            if done is not None:
                return done
            att = folded_edge_phase(sq[:, c:2 * c], sq[:, 2 * c:3 * c], sq[:, 3 * c:4 * c], sq[:, :c], sq[:, 4 * c:],
                                    edge_attr_csr, plan, self.num_heads, up, ld_out=wpf.shape[1])
            # projection(out + x_r) + x_skip, lin_edge part via W_t; the statistics of the MLP's LayerNorm ride on the
            # epilogue, those of the next block's layer_norm1 on the MLP's last Linear
            y = ops.linear(att, wpf, bp, residual=x, stats_eps=self._mlp_ln_eps("dst", dtype))
            return self._node_mlp(y, "dst", 1, out_stats_eps=self._next_ln_eps(dtype))
        wp, bp = self._cat_linear("proj", [self.projection], dtype)
        we, be = self._edge_params()
        sqkv = self._ln_linear(xh, "sqkv", lambda: self._cat_linear("sqkv", all4, dtype), lambda: self._cat_rows(all4),
                               [l.weight for l in all4] + [l.bias for l in all4])  # [N, 4C] = x_r | q | k | v
        att = self.conv.fused(sqkv[:, c:2 * c], sqkv[:, 2 * c:3 * c], sqkv[:, 3 * c:], sqkv[:, :c], edge_attr_csr,
                              self.edge_dim, we, be, plan, self.num_heads)
        y = ops.linear(att, wp, bp, residual=x, stats_eps=self._mlp_ln_eps("dst", dtype))  # projection(out + x_r) + x
        return self._node_mlp(y, "dst", 1, out_stats_eps=self._next_ln_eps(dtype))

    def _sharded(self, x: Tensor, edge_attr: Tensor, edge_index: Tensor, shapes: tuple, batch_size: int, model_comm_group,
                 size=None) -> Tensor:
        """The reference's module-level protocol across a model group (layers/block.py:602-635 with ``shard_qkve_heads`` /
        ``shard_output_seq``): ``x`` and ``edge_attr`` are this rank's row / edge shards, ``edge_index`` the whole edge
        list.  Everything is row-local except the conv, which runs on all nodes and edges for this rank's heads
        (``GraphTransformerConv.forward`` = ``anemoi_gt_conv``).  One code path on the autograd nodes (they run their
        forward kernels under ``no_grad`` as well); the model root partitions by mesh node instead and is the fast route."""
        from .. import autograd

        dtype = runtime.compute_dtype(x)
        x = _as_compute(x, dtype)
        ln = self.layer_norm1
        h = autograd.layer_norm(x, ln.weight, ln.bias, ln.eps)
        x_r, q, k, v = (autograd.linear(h, lin.weight, lin.bias) for lin in (self.lin_self, self.lin_query, self.lin_key,
                                                                              self.lin_value))
        e = autograd.linear(_as_compute(edge_attr, dtype), self.lin_edge.weight, self.lin_edge.bias)
        q, k, v, e = self.shard_qkve_heads(q, k, v, e, shapes, batch_size, model_comm_group)
        out = self.conv(q, k, v, e, edge_index, size=size)
        out = self.shard_output_seq(out, shapes, batch_size, model_comm_group)
        out = autograd.linear(out + x_r, self.projection.weight, self.projection.bias, "Identity", x)
        return training.sequential(self.node_dst_mlp, out, residual=out)

    def forward(
        self,
        x: Tensor,
        edge_attr: Tensor,
        edge_index: Tensor,
        shapes: tuple,
        batch_size: int,
        model_comm_group=None,
        size=None,
    ):
        if _group_size(model_comm_group) > 1:
            assert batch_size == 1, "Only batch size of 1 is supported when model is sharded across GPUs"
            return self._sharded(x, edge_attr, edge_index, shapes, batch_size, model_comm_group, size), edge_attr
        if training.wants_grad(self, x, edge_attr):
            return training.gt_processor_block(self, x, edge_attr, edge_index, size)
        dtype = runtime.compute_dtype(x)
        n = x.shape[0]
        if size is not None and tuple(size) != (n, n):
            raise ValueError(f"Encountered tensor with size {n} in dimension 0, but expected size {tuple(size)}")
        plan, ea = self._edge_inputs(edge_attr, edge_index, n, n, dtype)
        return self.native(_as_compute(x, dtype), ea, plan), edge_attr


class GraphTransformerMapperBlock(GraphTransformerBaseBlock):
    """Graph transformer layer between two node sets (reference layers/block.py:429-550)."""

    def __init__(
        self,
        in_channels: int,
        hidden_dim: int,
        out_channels: int,
        edge_dim: int,
        num_heads: int = 16,
        bias: bool = True,
        activation: str = "GELU",
        num_chunks: int = 1,
        update_src_nodes: bool = False,
        **kwargs,
    ) -> None:
        super().__init__(
            in_channels=in_channels, hidden_dim=hidden_dim, out_channels=out_channels, edge_dim=edge_dim,
            num_heads=num_heads, bias=bias, activation=activation, num_chunks=num_chunks,
            update_src_nodes=update_src_nodes, **kwargs,
        )
        self.layer_norm2 = nn.LayerNorm(in_channels)

    def native(self, x_src: Tensor, x_dst: Tensor, edge_attr_csr: Tensor, plan: EdgePlan, num_chunks: int = 1,
               halo=None, out_stats_eps: Optional[float] = None):
        """``out_stats_eps``: epsilon of the LayerNorm the new destination nodes enter next (its statistics then come
        out of the node MLP's last GEMM).  ``halo``: node-partitioned run -- ``x_src`` holds this rank's source rows, the k|v rows of the other
        sources of the plan are appended by one all-to-all-v."""
        dtype = x_dst.dtype
        self._check_channels(dtype)
        c = self.num_heads * self.out_channels_conv
        # EmbeddedRows (mappers): the rows arrive as the raw features they are embedded from; ``h_dst`` is materialised
        h_dst = x_dst.h if isinstance(x_dst, EmbeddedRows) else x_dst
        if isinstance(x_src, EmbeddedRows) and x_src.h is None and self.update_src_nodes:
            raise ValueError("update_src_nodes needs the embedded source rows")
        kv_layers, sq_layers = [self.lin_key, self.lin_value], [self.lin_self, self.lin_query]
        kv_args = ("kv", lambda: self._cat_linear("kv", kv_layers, dtype), lambda: self._cat_rows(kv_layers),
                   [l.weight for l in kv_layers] + [l.bias for l in kv_layers])
        xs = self._ln_begin(self.layer_norm1, x_src)
        if halo is None:
            kv = self._ln_linear(xs, *kv_args)  # [N_src, 2C] = k | v
        else:
            n_own = x_src.shape[0]
            kv = torch.empty((n_own + halo.n_recv, 2 * c), dtype=dtype, device=x_src.device)
            self._ln_linear(xs, *kv_args, out=kv[:n_own])
            pending = halo.start(kv, n_own)  # overlapped with the destination-side LayerNorm + GEMM below
        del xs
        xd = self._ln_begin(self.layer_norm2, x_dst)
        up = self.fold_width(dtype)
        if up is not None:
            wp, bp = self._folded_out(dtype, up)
            sq = self._ln_linear(xd, "squ", lambda: self._folded_in("squ", sq_layers, dtype, up),
                                 lambda: self._folded_rows(sq_layers, up),
                                 self._fold_params(sq_layers))  # [N_dst, 2C + H*up] = x_r | q | u
            del xd
            if halo is not None:
                halo.finish(pending)
            if num_chunks <= 1 and not self.update_src_nodes:
                done = self._block_tail(sq[:, c:2 * c], kv[:, :c], kv[:, c:], sq[:, :c], sq[:, 2 * c:], edge_attr_csr, plan,
                                        h_dst, up, out_stats_eps)
                if done is not None:
                    return (x_src.h if isinstance(x_src, EmbeddedRows) else x_src), done
            att = folded_edge_phase(sq[:, c:2 * c], kv[:, :c], kv[:, c:], sq[:, :c], sq[:, 2 * c:], edge_attr_csr, plan,
                                    self.num_heads, up, ld_out=wp.shape[1])
        else:
            if halo is not None:
                raise NotImplementedError("node-partitioned blocks need the folded edge kernel")
            wp, bp = self._cat_linear("proj", [self.projection], dtype)
            we, be = self._edge_params()
            sq = self._ln_linear(xd, "sq", lambda: self._cat_linear("sq", sq_layers, dtype),
                                 lambda: self._cat_rows(sq_layers),
                                 [l.weight for l in sq_layers] + [l.bias for l in sq_layers])  # [N_dst, 2C] = x_r | q
            del xd
            att = self.conv.fused(sq[:, c:], kv[:, :c], kv[:, c:], sq[:, :c], edge_attr_csr, self.edge_dim, we, be,
                                  plan, self.num_heads)
        del sq, kv
        y = ops.linear(att, wp, bp, residual=h_dst,
                       stats_eps=self._mlp_ln_eps("dst", dtype) if num_chunks <= 1 else None)
        del att
        new_dst = self._node_mlp(y, "dst", num_chunks, out_stats_eps=out_stats_eps)
        if isinstance(x_src, EmbeddedRows):
            x_src = x_src.h
        new_src = self._node_mlp(x_src, "src", num_chunks) if self.update_src_nodes else x_src
        return new_src, new_dst

    def _sharded(self, x, edge_attr: Tensor, edge_index: Tensor, shapes: tuple, batch_size: int, model_comm_group, size=None):
        """The reference's module-level protocol across a model group (layers/block.py:479-550): row shards of the source
        and destination nodes, an edge shard of the attributes, the whole edge index; heads exchanged around the conv
        (see ``GraphTransformerProcessorBlock._sharded``)."""
        from .. import autograd

        x_src, x_dst = x
        dtype = runtime.compute_dtype(x_dst)
        x_src, x_dst = _as_compute(x_src, dtype), _as_compute(x_dst, dtype)
        ln1, ln2 = self.layer_norm1, self.layer_norm2
        hs = autograd.layer_norm(x_src, ln1.weight, ln1.bias, ln1.eps)
        hd = autograd.layer_norm(x_dst, ln2.weight, ln2.bias, ln2.eps)
        x_r = autograd.linear(hd, self.lin_self.weight, self.lin_self.bias)
        q = autograd.linear(hd, self.lin_query.weight, self.lin_query.bias)
        k = autograd.linear(hs, self.lin_key.weight, self.lin_key.bias)
        v = autograd.linear(hs, self.lin_value.weight, self.lin_value.bias)
        e = autograd.linear(_as_compute(edge_attr, dtype), self.lin_edge.weight, self.lin_edge.bias)
        q, k, v, e = self.shard_qkve_heads(q, k, v, e, shapes, batch_size, model_comm_group)
        out = self.conv(q, k, v, e, edge_index, size=size)
        out = self.shard_output_seq(out, shapes, batch_size, model_comm_group)
        out = autograd.linear(out + x_r, self.projection.weight, self.projection.bias, "Identity", x_dst)
        # update_src_nodes: row-local on whatever source rows this rank holds (reference layers/block.py:540-546)
        new_src = training.sequential(self.node_src_mlp, x_src, residual=x_src) if self.update_src_nodes else x_src
        return new_src, training.sequential(self.node_dst_mlp, out, residual=out)

    def forward(
        self,
        x,
        edge_attr: Tensor,
        edge_index: Tensor,
        shapes: tuple,
        batch_size: int,
        model_comm_group=None,
        size=None,
    ):
        if _group_size(model_comm_group) > 1:
            assert batch_size == 1, "Only batch size of 1 is supported when model is sharded across GPUs"
            return self._sharded(x, edge_attr, edge_index, shapes, batch_size, model_comm_group, size), edge_attr
        if training.wants_grad(self, x[0], x[1], edge_attr):
            return training.gt_mapper_block(self, x, edge_attr, edge_index, size)
        x_src, x_dst = x
        dtype = runtime.compute_dtype(x_dst)
        n_src, n_dst = x_src.shape[0], x_dst.shape[0]
        if size is not None and tuple(size) != (n_src, n_dst):
            raise ValueError(f"Encountered tensors with sizes {(n_src, n_dst)}, but expected size {tuple(size)}")
        plan, ea = self._edge_inputs(edge_attr, edge_index, n_src, n_dst, dtype)
        num_chunks = self.num_chunks if self.training else inference_num_chunks()
        new_src, new_dst = self.native(_as_compute(x_src, dtype), _as_compute(x_dst, dtype), ea, plan, num_chunks)
        return (new_src if self.update_src_nodes else x[0], new_dst), edge_attr


# =============================================================================================
# GNN (edge-MLP message passing) blocks -- parameters mirror reference layers/block.py:108-286
# =============================================================================================
class GraphConvBaseBlock(BaseBlock, ABC):
    def __init__(
        self,
        in_channels: int,
        out_channels: int,
        mlp_extra_layers: int = 0,
        activation: str = "SiLU",
        update_src_nodes: bool = True,
        num_chunks: int = 1,
        **kwargs,
    ) -> None:
        super().__init__(**kwargs)
        self.update_src_nodes = update_src_nodes
        self.num_chunks = num_chunks
        self.node_mlp = MLP(2 * in_channels, out_channels, out_channels, n_extra_layers=mlp_extra_layers,
                            activation=activation)
        self.conv = GraphConv(in_channels=in_channels, out_channels=out_channels, mlp_extra_layers=mlp_extra_layers,
                              activation=activation)
        self._packed = runtime.PackedWeights()
        self._plans = runtime.PlanCache()

    @abstractmethod
    def forward(self, x, edge_attr: Tensor, edge_index: Tensor, shapes: tuple, model_comm_group=None, size=None):
        """``(new nodes, new edge state)`` -- reference layers/block.py:157-167."""

    @staticmethod
    def _check_width(width: int, dtype) -> None:
        """The node / edge width splits the first edge-MLP Linear into column blocks that are GEMM operands of their own
        (K = width): it has to be a whole number of the GEMM's K slabs, as the graph-transformer blocks' hidden width."""
        mult = ops.k_multiple(dtype)
        if width % mult != 0:
            raise NotImplementedError(f"hidden width {width} must be a multiple of {mult} for {dtype} on the MI355X path")


class GraphConvProcessorBlock(GraphConvBaseBlock):
    """Edge-MLP message passing on one node set (reference layers/block.py:170-223, layers/conv.py:27-76)."""

    def native(self, x: Tensor, e_csr: Tensor, plan: EdgePlan, halo=None):
        """x ``[N, C]``, edge state ``[E, C]`` in CSR (destination-sorted) order -> (new nodes, new edge state).

        ``halo`` (node-partitioned run): ``x`` holds this rank's rows; the ``W1b x`` rows of halo sources are fetched
        from their owners by one all-to-all-v (C values per halo node) while the edge-side GEMM runs."""
        dtype = x.dtype
        c = x.shape[1]
        self._check_width(c, dtype)
        edge_mlp, node_mlp = self.conv.edge_mlp.native(), self.node_mlp.native()
        lin1 = edge_mlp.steps[0][1]
        act1 = edge_mlp.steps[0][2]
        if lin1.in_features != 3 * c or e_csr.shape[1] != c:
            raise ValueError(f"GNN block expects node / edge width {lin1.in_features // 3}, got {c} / {e_csr.shape[1]}")
        if e_csr.shape[0] == 0 and halo is None:  # an edge set without edges: empty new state, zero sums, nothing launched on it
            xcat = torch.zeros((x.shape[0], 2 * c), dtype=dtype, device=x.device)
            xcat[:, :c].copy_(x)
            return node_mlp(xcat, residual=x), e_csr
        # W1 [x_i | x_j | e] = (W1a x)_i + (W1b x)_j + W1c e : node part as ONE [N, 2C] GEMM, edge part as [E, C] GEMM
        w_nodes = self._packed.get(("w1_nodes", dtype), [lin1.weight],
                                   lambda: runtime.pack_weight([lin1.weight[:, :c], lin1.weight[:, c:2 * c]], dtype))
        w_edges = self._packed.get(("w1_edges", dtype), [lin1.weight],
                                   lambda: runtime.pack_weight([lin1.weight[:, 2 * c:]], dtype))
        b1 = None if lin1.bias is None else runtime.f32c(lin1.bias)
        if halo is None:
            p = ops.linear(x, w_nodes, None)  # [N, 2C] = W1a x | W1b x
            p_dst, p_src = p[:, :c], p[:, c:]
            t = ops.linear(e_csr, w_edges, b1)
        else:
            n_own = x.shape[0]
            p_src = torch.empty((n_own + halo.n_recv, c), dtype=dtype, device=x.device)
            ops.linear(x, w_nodes[c:], None, out=p_src[:n_own])  # W1b x of the own rows, halo rows appended below
            pending = halo.start(p_src, n_own)
            p_dst = ops.linear(x, w_nodes[:c], None)
            t = ops.linear(e_csr, w_edges, b1)
            halo.finish(pending)
            p = None
        h = ops.gather_add_act(t, p_dst, p_src, plan.dst, plan.col, act=act1, out=t)
        e_new = edge_mlp(h, residual=e_csr, start=1)  # remaining Linear/act pairs, LayerNorm, "+ e"
        del h, t, p, p_dst, p_src
        xcat = ops.segment_sum(e_new, plan.rowptr, cat_with=x)  # [x | scatter-sum over destinations]: the node MLP's input
        return node_mlp(xcat, residual=x), e_new

    def _sharded(self, x, edge_attr, edge_index, shapes, model_comm_group, size=None):
        """The reference's module-level protocol across a model group (layers/block.py:193-223): ``x`` is this rank's row
        shard, ``edge_attr`` / ``edge_index`` its 1-hop edge shard (global node ids).  ``sync_tensor`` gathers the nodes,
        the conv updates the local edges and sums them over ALL destinations, ``shard_tensor`` keeps this rank's rows."""
        from ..distributed.graph import shard_tensor, sync_tensor

        dtype = runtime.compute_dtype(x)
        x = _as_compute(x, dtype)
        x_in = sync_tensor(x, 0, shapes[1], model_comm_group)
        out, edges_new = self.conv(x_in, _as_compute(edge_attr, dtype), edge_index, size=size)
        out = shard_tensor(out, 0, shapes[1], model_comm_group, gather_in_backward=False)
        return training.mlp(self.node_mlp, torch.cat([x, out], dim=1), residual=x), edges_new

    def forward(self, x, edge_attr, edge_index, shapes, model_comm_group=None, size=None):
        if _group_size(model_comm_group) > 1:
            return self._sharded(x, edge_attr, edge_index, shapes, model_comm_group, size)
        if training.wants_grad(self, x, edge_attr):
            return training.gnn_processor_block(self, x, edge_attr, edge_index, size)
        dtype = runtime.compute_dtype(x)
        n = x.shape[0]
        plan = self._plans.get(edge_index, n, n)
        perm = plan.perm.long()
        e_csr = _as_compute(edge_attr, dtype).index_select(0, perm)
        x_new, e_new = self.native(_as_compute(x, dtype), e_csr, plan)
        edges_new = torch.empty_like(e_new)
        edges_new[perm] = e_new  # back to the caller's edge order
        return x_new, edges_new


class GraphConvMapperBlock(GraphConvBaseBlock):
    """Edge-MLP message passing between two node sets (reference layers/block.py:226-286)."""

    def native(self, x_src: Tensor, x_dst: Tensor, e_csr: Tensor, plan: EdgePlan, halo=None):
        """``halo`` (node-partitioned run, backward mapper): ``x_src`` holds this rank's source rows only; the ``W1b x``
        rows of the halo sources arrive from their owners by one all-to-all-v (C values per row) while the destination
        and edge GEMMs run."""
        dtype = x_dst.dtype
        c = x_dst.shape[1]
        self._check_width(c, dtype)
        edge_mlp, node_mlp = self.conv.edge_mlp.native(), self.node_mlp.native()
        lin1, act1 = edge_mlp.steps[0][1], edge_mlp.steps[0][2]
        if e_csr.shape[0] == 0 and halo is None:  # (see GraphConvProcessorBlock.native)
            def lone(x, second):  # node_mlp(cat[x, second]) + x
                xcat = torch.zeros((x.shape[0], 2 * c), dtype=dtype, device=x.device)
                xcat[:, :c].copy_(x)
                if second is not None:
                    xcat[:, c:].copy_(second)
                return node_mlp(xcat, residual=x)

            return (lone(x_src, x_src) if self.update_src_nodes else x_src, lone(x_dst, None)), e_csr
        w_dst = self._packed.get(("w1_dst", dtype), [lin1.weight], lambda: runtime.pack_weight([lin1.weight[:, :c]], dtype))
        w_src = self._packed.get(("w1_src", dtype), [lin1.weight],
                                 lambda: runtime.pack_weight([lin1.weight[:, c:2 * c]], dtype))
        w_edges = self._packed.get(("w1_edges", dtype), [lin1.weight],
                                   lambda: runtime.pack_weight([lin1.weight[:, 2 * c:]], dtype))
        if halo is None:
            p_src = ops.linear(x_src, w_src, None)
            pending = None
        else:
            n_own = x_src.shape[0]
            p_src = torch.empty((n_own + halo.n_recv, c), dtype=dtype, device=x_dst.device)
            ops.linear(x_src, w_src, None, out=p_src[:n_own])
            pending = halo.start(p_src, n_own)
        p_dst = ops.linear(x_dst, w_dst, None)
        t = ops.linear(e_csr, w_edges, None if lin1.bias is None else runtime.f32c(lin1.bias))
        if pending is not None:
            halo.finish(pending)
        h = ops.gather_add_act(t, p_dst, p_src, plan.dst, plan.col, act=act1, out=t)
        e_new = edge_mlp(h, residual=e_csr, start=1)
        del h, t, p_dst, p_src

        def update(x, agg):  # node_mlp(cat[x, agg]) + x
            if agg is not None:
                return node_mlp(ops.segment_sum(agg, plan.rowptr, cat_with=x), residual=x)
            xcat = torch.empty((x.shape[0], 2 * c), dtype=dtype, device=x.device)
            xcat[:, :c].copy_(x)
            xcat[:, c:].copy_(x)
            return node_mlp(xcat, residual=x)

        new_dst = update(x_dst, e_new)
        new_src = update(x_src, None) if self.update_src_nodes else x_src  # reference block.py:282
        return (new_src, new_dst), e_new

    def _sharded(self, x, edge_attr, edge_index, shapes, model_comm_group, size=None):
        """Reference layers/block.py:249-286 across a model group: both node sets gathered (``sync_tensor``), the conv on
        the local 1-hop edges, this rank's destination rows kept; the source update is row-local."""
        from ..distributed.graph import shard_tensor, sync_tensor

        dtype = runtime.compute_dtype(x[1])
        x_src, x_dst = _as_compute(x[0], dtype), _as_compute(x[1], dtype)
        x_in = (sync_tensor(x_src, 0, shapes[0], model_comm_group), sync_tensor(x_dst, 0, shapes[1], model_comm_group))
        out, edges_new = self.conv(x_in, _as_compute(edge_attr, dtype), edge_index, size=size)
        out = shard_tensor(out, 0, shapes[1], model_comm_group, gather_in_backward=False)
        new_dst = training.mlp(self.node_mlp, torch.cat([x_dst, out], dim=1), residual=x_dst)
        new_src = x_src if not self.update_src_nodes else training.mlp(self.node_mlp, torch.cat([x_src, x_src], dim=1),
                                                                      residual=x_src)
        return (new_src, new_dst), edges_new

    def forward(self, x, edge_attr, edge_index, shapes, model_comm_group=None, size=None):
        if _group_size(model_comm_group) > 1:
            return self._sharded(x, edge_attr, edge_index, shapes, model_comm_group, size)
        if training.wants_grad(self, x[0], x[1], edge_attr):
            return training.gnn_mapper_block(self, x, edge_attr, edge_index, size)
        x_src, x_dst = x
        dtype = runtime.compute_dtype(x_dst)
        n_src, n_dst = x_src.shape[0], x_dst.shape[0]
        if size is not None and tuple(size) != (n_src, n_dst):
            raise ValueError(f"Encountered tensors with sizes {(n_src, n_dst)}, but expected size {tuple(size)}")
        plan = self._plans.get(edge_index, n_src, n_dst)
        perm = plan.perm.long()
        e_csr = _as_compute(edge_attr, dtype).index_select(0, perm)
        nodes, e_new = self.native(_as_compute(x_src, dtype), _as_compute(x_dst, dtype), e_csr, plan)
        edges_new = torch.empty_like(e_new)
        edges_new[perm] = e_new
        return nodes, edges_new


# =============================================================================================
# Transformer block (mesh-node multi-head self attention) -- reference layers/block.py:61-105
# =============================================================================================
class TransformerProcessorBlock(BaseBlock):
    def __init__(self, num_channels: int, hidden_dim: int, num_heads: int, activation: str, window_size: int,
                 dropout_p: float = 0.0):
        super().__init__()
        from .attention import MultiHeadSelfAttention

        act = activation_class(activation)
        self.layer_norm1 = nn.LayerNorm(num_channels)
        self.attention = MultiHeadSelfAttention(num_heads=num_heads, embed_dim=num_channels, window_size=window_size,
                                                bias=False, is_causal=False, dropout_p=dropout_p)
        self.mlp = nn.Sequential(nn.Linear(num_channels, hidden_dim), act(), nn.Linear(hidden_dim, num_channels))
        self.layer_norm2 = nn.LayerNorm(num_channels)

        self._mlp: Optional[NativeSequential] = None

    def native(self, x: Tensor, batch_size: int, head_exchange=None) -> Tensor:
        """Pre-LN attention residual + pre-LN MLP residual (reference layers/block.py:99-105).

        ``head_exchange`` (node-partitioned run): q|k|v of the own rows go through an all-to-all so that this rank holds
        ALL rows of its share of the heads, attention runs on those heads, and a second all-to-all brings the own rows
        of all heads back."""
        ln1, ln2 = self.layer_norm1, self.layer_norm2
        att = self.attention
        if head_exchange is None:
            done = self._block_abi(x, batch_size)
            if done is not None:
                return done
        h = ops.layer_norm(x, runtime.f32c(ln1.weight), runtime.f32c(ln1.bias), ln1.eps)
        qkv = linear_native(att._packed, "lin_qkv", att.lin_qkv, h)
        # attention dropout (training mode, reference layers/attention.py:90) across a model group: ONE seed -- rank 0's draw,
        # broadcast -- and the GLOBAL head index in the mask's hash, so the ranks together drop exactly what the unsharded
        # attention drops with that seed (every rank sees the whole sequence of its heads in the unsharded row order)
        drop_p, drop_seed, drop_dev = att.dropout(None if head_exchange is None else head_exchange.group)
        if head_exchange is not None:
            window = att.attention_window()
            qkv_heads = head_exchange.rows_to_heads(qkv, att.num_heads)  # [S, 3 * C_local], internal row order
            if window >= 0:  # the window slides over the EXTERNAL node order
                qkv_heads = qkv_heads.index_select(0, head_exchange.to_external)
            a_heads = ops.mhsa(qkv_heads, batch_size, head_exchange.local_heads(att.num_heads), window, dropout_p=drop_p,
                               dropout_seed=drop_seed, head_offset=head_exchange._head_bounds(att.num_heads)[head_exchange.rank],
                               heads_total=att.num_heads, seed_dev=drop_dev)
            if window >= 0:
                a_heads = a_heads.index_select(0, head_exchange.to_internal)
            a = head_exchange.heads_to_rows(a_heads, att.num_heads)  # [n_own, C]
        else:
            a = ops.mhsa(qkv, batch_size, att.num_heads, att.attention_window(), dropout_p=drop_p, dropout_seed=drop_seed,
                         seed_dev=drop_dev)
        x = linear_native(att._packed, "projection", att.projection, a, residual=x)  # x + attention(...)
        if self._mlp is None:
            self._mlp = NativeSequential(self.mlp)
        h = ops.layer_norm(x, runtime.f32c(ln2.weight), runtime.f32c(ln2.bias), ln2.eps)
        return self._mlp(h, residual=x)

    block_abi = True  # False: op by op (tests compare the two routes)

    def _block_abi(self, x: Tensor, batch_size: int) -> Optional[Tensor]:
        """The whole block as ONE call of ``anemoi_transformer_block_forward`` (the launches and packed weights of the
        op-by-op route below; bit-identical), f32 or bf16 -- or ``None`` when this call has to go op by op (bench.py's
        per-kernel timing pass, channel widths that need K padding, an MLP of another shape)."""
        import ctypes

        from .. import _lib

        dtype, att = x.dtype, self.attention
        c, rows = x.shape[1], x.shape[0]
        mult = ops.k_multiple(dtype)
        if (not self.block_abi or ops.PROFILE is not None or not x.is_cuda or rows == 0 or x.stride(1) != 1
                or c != att.embed_dim or c % mult != 0 or rows % batch_size != 0):
            return None
        if self._mlp is None:
            self._mlp = NativeSequential(self.mlp)
        steps = self._mlp.steps
        if [k for k, _, _ in steps] != ["linear", "linear"] or steps[0][2] not in _lib.ACT_CODES or steps[1][2] != "Identity":
            return None
        fc1, fc2 = steps[0][1], steps[1][1]
        hidden = fc1.out_features
        if hidden % mult != 0 or fc1.in_features != c or fc2.out_features != c:
            return None

        def packed(cache, tag, lin):
            w = cache.get((tag, "w", dtype), [lin.weight], lambda: runtime.pack_weight([lin.weight], dtype))
            return w, (None if lin.bias is None else runtime.f32c(lin.bias))

        w_qkv, b_qkv = packed(att._packed, "lin_qkv", att.lin_qkv)
        w_proj, b_proj = packed(att._packed, "projection", att.projection)
        w1 = self._mlp.cache.get(("w", 0, dtype), [fc1.weight], lambda: runtime.pack_weight([fc1.weight], dtype))
        w2 = self._mlp.cache.get(("w", 1, dtype), [fc2.weight], lambda: runtime.pack_weight([fc2.weight], dtype))
        b1 = None if fc1.bias is None else runtime.f32c(fc1.bias)
        b2 = None if fc2.bias is None else runtime.f32c(fc2.bias)
        ln1, ln2 = self.layer_norm1, self.layer_norm2
        p, seed, seed_dev = att.dropout()
        new = lambda *shape: torch.empty(shape, dtype=dtype, device=x.device)  # noqa: E731
        h_ln, qkv, a_out, y, hid, out = new(rows, c), new(rows, 3 * c), new(rows, c), new(rows, c), new(rows, hidden), new(rows, c)
        lib = _lib.load()
        s_len, heads = rows // batch_size, att.num_heads
        ws_bytes = lib.anemoi_mhsa_workspace_bytes(ops.dtype_code(dtype), batch_size, s_len, heads, c // heads)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device) if ws_bytes > 0 else None
        keep = [runtime.f32c(t) for t in (ln1.weight, ln1.bias, ln2.weight, ln2.bias)]
        a = _lib.TfmBlockArgs()
        a.struct_bytes, a.rows, a.dtype = ctypes.sizeof(_lib.TfmBlockArgs), rows, ops.dtype_code(dtype)
        a.B, a.S, a.C, a.H, a.hidden, a.act = batch_size, s_len, c, heads, hidden, _lib.ACT_CODES[steps[0][2]]
        a.window, a.eps1, a.eps2 = att.attention_window(), ln1.eps, ln2.eps
        a.dropout_p, a.dropout_seed, a.dropout_seed_dev = float(p), int(seed) & 0xFFFFFFFF, ops._seed_dev_ptr(seed_dev, x)
        a.dropout_h0, a.dropout_h_total = 0, 0
        a.x, a.ldx = x.data_ptr(), ops._ld(x)
        a.ln1_w, a.ln1_b, a.ln2_w, a.ln2_b = (t.data_ptr() for t in keep)
        a.w_qkv, a.b_qkv, a.w_proj, a.b_proj = w_qkv.data_ptr(), ops._ptr(b_qkv), w_proj.data_ptr(), ops._ptr(b_proj)
        a.w_fc1, a.b_fc1, a.w_fc2, a.b_fc2 = w1.data_ptr(), ops._ptr(b1), w2.data_ptr(), ops._ptr(b2)
        a.h_ln, a.qkv, a.att, a.y, a.h = h_ln.data_ptr(), qkv.data_ptr(), a_out.data_ptr(), y.data_ptr(), hid.data_ptr()
        a.mhsa_ws, a.out = ops._ptr(ws), out.data_ptr()
        _lib.check(lib.anemoi_transformer_block_forward(ctypes.byref(a), ops._stream()), "anemoi_transformer_block_forward")
        return out

    def _sharded(self, x: Tensor, shapes: list, batch_size: int, model_comm_group) -> Tensor:
        """Sequence-sharded call as in the reference (layers/block.py:99-105 with a model group): everything but the
        attention is row-local; the attention module reshards rows <-> heads around its kernel."""
        from .. import autograd

        grad = training.wants_grad(self, x)
        dtype = runtime.compute_dtype(x)
        x = _as_compute(x, dtype)
        ln1, ln2 = self.layer_norm1, self.layer_norm2
        if grad:
            h = autograd.layer_norm(x, ln1.weight, ln1.bias, ln1.eps)
            x = x + self.attention(h, shapes, batch_size, model_comm_group)
            h = autograd.layer_norm(x, ln2.weight, ln2.bias, ln2.eps)
            return training.sequential(self.mlp, h, residual=x)
        h = ops.layer_norm(x, runtime.f32c(ln1.weight), runtime.f32c(ln1.bias), ln1.eps)
        x = ops.add(x, self.attention(h, shapes, batch_size, model_comm_group))
        if self._mlp is None:
            self._mlp = NativeSequential(self.mlp)
        h = ops.layer_norm(x, runtime.f32c(ln2.weight), runtime.f32c(ln2.bias), ln2.eps)
        return self._mlp(h, residual=x)

    def forward(self, x: Tensor, shapes: list, batch_size: int, model_comm_group=None) -> Tensor:
        if _group_size(model_comm_group) > 1:
            return self._sharded(x, shapes, batch_size, model_comm_group)
        if training.wants_grad(self, x):
            return training.transformer_block(self, x, batch_size)
        dtype = runtime.compute_dtype(x)
        return self.native(_as_compute(x, dtype), batch_size)
