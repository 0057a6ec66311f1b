"""Grid <-> mesh mappers mirroring reference layers/mapper.py.

``GraphTransformerForwardMapper`` (data -> hidden) and ``GraphTransformerBackwardMapper`` (hidden -> data):
node embeddings on the fused GEMM, one :class:`GraphTransformerMapperBlock` on the bipartite CSR plan, and for
the backward mapper the LayerNorm + Linear extractor.  Sub-module / buffer names match the reference, so a
reference ``state_dict`` loads unchanged (``edge_inc`` persistent; ``edge_attr`` / ``edge_index_base`` not).
"""

from __future__ import annotations

from abc import ABC
from typing import Optional

import numpy as np
import torch
from torch import Tensor
from torch import nn

from .. import ops
from .. import runtime
from .. import training
from .block import EmbeddedRows
from .block import GraphConvMapperBlock
from .block import GraphTransformerMapperBlock
from .block import inference_num_chunks
from .graph import TrainableTensor
from .mlp import MLP
from .mlp import linear_native


class BaseMapper(nn.Module, ABC):
    def __init__(self, in_channels_src: int = 0, in_channels_dst: int = 0, hidden_dim: int = 128,
                 out_channels_dst: Optional[int] = None, cpu_offload: bool = False, activation: str = "SiLU",
                 **kwargs) -> None:
        super().__init__()
        self.in_channels_src = in_channels_src
        self.in_channels_dst = in_channels_dst
        self.hidden_dim = hidden_dim
        self.out_channels_dst = out_channels_dst
        self.activation = activation
        self.proc = NotImplemented
        self.offload_layers(cpu_offload)

    def offload_layers(self, cpu_offload: bool) -> None:
        """``cpu_offload=True`` (reference: ``offload_wrapper`` around the layers, layers/mapper.py:64-66 /
        layers/processor.py:65-67 -- torch's ``OffloadWrapper`` = ``save_on_cpu``): the tensors the differentiable route saves
        for its backward live in pinned host memory between forward and backward.  Inference is unaffected.  Not under HIP
        graph capture (the copies synchronise)."""
        self.cpu_offload = bool(cpu_offload)

    def _offloaded(self):
        import contextlib

        if getattr(self, "cpu_offload", False):
            return torch.autograd.graph.save_on_cpu(pin_memory=True)
        return contextlib.nullcontext()

    # ---- the reference's pre / post processing hooks (layers/mapper.py:68-116, 412-418, 690-694) --------------------
    # ``forward`` runs fused launch sequences and does not come through these; they are kept, with the reference's
    # signatures and return values, for callers (and the reference's own tests) that use a mapper piecewise.
    @staticmethod
    def _apply_module(module: nn.Module, x: Tensor) -> Tensor:
        """An embedding / extraction sub-module on node rows, on the HIP kernels with or without an autograd graph."""
        from .mlp import MLP, NativeSequential

        if isinstance(module, MLP):
            return module(x)
        dtype = runtime.compute_dtype(x)
        seq = module if isinstance(module, nn.Sequential) else nn.Sequential(module)
        if training.wants_grad(seq, x):
            return training.sequential(seq, training._cast(x, dtype))
        xin = x if x.dtype == dtype else x.to(dtype)
        return NativeSequential(seq)(xin if xin.stride(-1) == 1 else xin.contiguous())

    def pre_process(self, x, shard_shapes, model_comm_group=None):
        """``(x_src, x_dst, shapes_src, shapes_dst)`` (reference layers/mapper.py:68-89)."""
        shapes_src, shapes_dst = shard_shapes
        x_src, x_dst = x
        return x_src, x_dst, shapes_src, shapes_dst

    def post_process(self, x_dst, shapes_dst, model_comm_group=None):
        return x_dst


class ForwardMapperPreProcessMixin:
    """data -> hidden: both node sets are embedded (reference layers/mapper.py:105-116)."""

    def pre_process(self, x, shard_shapes, model_comm_group=None):
        from ..distributed.shapes import change_channels_in_shape

        from ..distributed.graph import shard_tensor

        x_src, x_dst, shapes_src, shapes_dst = super().pre_process(x, shard_shapes, model_comm_group)
        x_src = shard_tensor(x_src, 0, shapes_src, model_comm_group)  # (identity without a model group)
        x_dst = shard_tensor(x_dst, 0, shapes_dst, model_comm_group)
        return (self._apply_module(self.emb_nodes_src, x_src), self._apply_module(self.emb_nodes_dst, x_dst),
                change_channels_in_shape(shapes_src, self.hidden_dim), change_channels_in_shape(shapes_dst, self.hidden_dim))


class BackwardMapperPostProcessMixin:
    """hidden -> data: the output variables are extracted (reference layers/mapper.py:96-102)."""

    def post_process(self, x_dst, shapes_dst, model_comm_group=None):
        from ..distributed.graph import gather_tensor
        from ..distributed.shapes import change_channels_in_shape

        x_dst = self._apply_module(self.node_data_extractor, x_dst)
        return gather_tensor(x_dst, 0, change_channels_in_shape(shapes_dst, self.out_channels_dst), model_comm_group)


class GraphEdgeMixin:
    """Edge buffers of a sub-graph (reference layers/mapper.py:119-171)."""

    def _register_edges(self, sub_graph, edge_attributes, src_size: int, dst_size: int, trainable_size: int) -> None:
        assert sub_graph, f"{self.__class__.__name__} needs a valid sub_graph to register edges."
        assert edge_attributes is not None, "Edge attributes must be provided"
        attr = torch.cat([sub_graph[name] for name in edge_attributes], dim=1)
        self.edge_dim = attr.shape[1] + trainable_size
        self.register_buffer("edge_attr", attr, persistent=False)
        self.register_buffer("edge_index_base", sub_graph.edge_index, persistent=False)
        self.register_buffer("edge_inc", torch.from_numpy(np.asarray([[src_size], [dst_size]], dtype=np.int64)),
                             persistent=True)

    def _expand_edges(self, edge_index: Tensor, edge_inc: Tensor, batch_size: int) -> Tensor:
        return runtime.expand_edges(edge_index, edge_inc, batch_size)


class GraphTransformerBaseMapper(GraphEdgeMixin, BaseMapper):
    def __init__(
        self,
        in_channels_src: int = 0,
        in_channels_dst: int = 0,
        hidden_dim: int = 128,
        trainable_size: int = 8,
        out_channels_dst: Optional[int] = None,
        num_chunks: int = 1,
        cpu_offload: bool = False,
        activation: str = "GELU",
        num_heads: int = 16,
        mlp_hidden_ratio: int = 4,
        sub_graph=None,
        sub_graph_edge_attributes: Optional[list] = None,
        src_grid_size: int = 0,
        dst_grid_size: int = 0,
    ) -> None:
        super().__init__(in_channels_src, in_channels_dst, hidden_dim, out_channels_dst=out_channels_dst,
                         num_chunks=num_chunks, cpu_offload=cpu_offload, activation=activation)
        self._register_edges(sub_graph, sub_graph_edge_attributes, src_grid_size, dst_grid_size, trainable_size)
        self.trainable = TrainableTensor(trainable_size=trainable_size, tensor_size=self.edge_attr.shape[0])
        self.proc = GraphTransformerMapperBlock(
            hidden_dim, mlp_hidden_ratio * hidden_dim, hidden_dim, num_heads=num_heads, edge_dim=self.edge_dim,
            activation=activation, num_chunks=num_chunks,
        )
        self.offload_layers(cpu_offload)
        self.emb_nodes_dst = nn.Linear(self.in_channels_dst, self.hidden_dim)
        self._packed = runtime.PackedWeights()
        self._plans = runtime.PlanCache()

    # ---- hooks specialised by the forward / backward mapper ------------------------------------
    def _embed(self, x_src: Tensor, x_dst: Tensor, one_cols=(None, None)):
        return x_src, x_dst  # the base mapper embeds nothing (reference BaseMapper.pre_process, layers/mapper.py:87-96)

    def _embedded(self, tag: str, lin: nn.Linear, x: Tensor, eps: Optional[float], one_col: Optional[int],
                  materialise: bool):
        """``lin(x)`` for the block -- as an ``EmbeddedRows`` handle when the block's LayerNorm -> Linear on these rows can
        run on the raw features instead (few input features, bf16 LayerNorm-fold path): ``x`` with a constant-1 column
        in its K padding (``one_col``: the caller already wrote it; None: an augmented copy is made here)."""
        k_in = lin.in_features
        if eps is None or not runtime.embed_fold_enabled(x.dtype) or 2 * (k_in + 1) > lin.out_features:
            return linear_native(self._packed, tag, lin, x, stats_eps=eps)
        if one_col is None:
            kp = ops.round_up(k_in + 1, ops.k_multiple(x.dtype))
            xa = torch.zeros((x.shape[0], kp), dtype=x.dtype, device=x.device)
            xa[:, :k_in].copy_(x[:, :k_in])
            xa[:, k_in].fill_(1.0)
            x, one_col = xa, k_in
        elif not (k_in <= one_col < x.shape[1]):
            raise ValueError(f"{tag}: constant-1 column {one_col} outside the K padding [{k_in}, {x.shape[1]})")
        h = linear_native(self._packed, tag, lin, x, stats_eps=eps) if materialise else None
        return EmbeddedRows(x, one_col, lin, h, self._packed, tag)

    def _extract(self, x_dst: Tensor, out_dtype) -> Tensor:
        return x_dst

    def _block_ln_eps(self, dtype):
        """(eps of the block's layer_norm1, layer_norm2) when their statistics can ride on the embedding GEMMs."""
        if not runtime.ln_fold_enabled(dtype):
            return None, None
        return self.proc.layer_norm1.eps, self.proc.layer_norm2.eps

    def _extract_ln_eps(self, dtype) -> Optional[float]:
        """eps of the LayerNorm that opens ``node_data_extractor`` (backward mapper), else None."""
        ext = getattr(self, "node_data_extractor", None)
        if isinstance(ext, nn.Sequential) and isinstance(ext[0], nn.LayerNorm) and runtime.ln_fold_enabled(dtype):
            return ext[0].eps
        return None

    def native(self, x_src: Tensor, x_dst: Tensor, batch_size: int, out_dtype: Optional[torch.dtype] = None,
               src_map: Optional[Tensor] = None, dst_map: Optional[Tensor] = None, one_cols=(None, None)) -> Tensor:
        """Inputs in the compute dtype (optionally K padded).  Returns the mapped destination nodes.

        ``src_map`` / ``dst_map``: optional external-id -> row relabelling when the caller keeps a node set in an
        internal order (the model root does this for the mesh).  ``one_cols = (src, dst)``: column of ``x_src`` /
        ``x_dst`` (inside the K padding) that already holds a constant 1 (``_embedded``), or None."""
        n_src, n_dst = x_src.shape[0], x_dst.shape[0]
        plan = self._plans.get(self.edge_index_base, n_src, n_dst, batch_size, self.edge_inc, src_map, dst_map)
        ea = runtime.edge_attr_csr_cached(self._packed, self.edge_attr, self.trainable.trainable, plan,
                                          *self.proc.edge_layout(x_dst.dtype))
        h_src, h_dst = self._embed(x_src, x_dst, one_cols)
        num_chunks = self.proc.num_chunks if self.training else inference_num_chunks()
        _, h_dst = self.proc.native(h_src, h_dst, ea, plan, num_chunks, out_stats_eps=self._extract_ln_eps(h_dst.dtype))
        return self._extract(h_dst, out_dtype)

    def native_local(self, x_src: Tensor, x_dst: Tensor, local_graph, out_dtype: Optional[torch.dtype] = None,
                     one_cols=(None, None)) -> Tensor:
        """Node-partitioned run (``distributed/partition.py``): local source / destination rows and a local CSR plan
        whose ``perm`` holds original edge ids; halo source rows (decoder) arrive by all-to-all-v inside the block."""
        plan = local_graph.plan
        ea = runtime.edge_attr_csr_cached(self._packed, self.edge_attr, self.trainable.trainable, plan,
                                          *self.proc.edge_layout(x_dst.dtype))
        h_src, h_dst = self._embed(x_src, x_dst, one_cols)
        num_chunks = self.proc.num_chunks if self.training else inference_num_chunks()
        _, h_dst = self.proc.native(h_src, h_dst, ea, plan, num_chunks, local_graph.halo,
                                    out_stats_eps=self._extract_ln_eps(h_dst.dtype))
        return self._extract(h_dst, out_dtype)

    def _run_sharded(self, x, batch_size: int, shard_shapes, model_comm_group) -> Tensor:
        """The reference's mapper forward across a model group (layers/mapper.py:239-272): the edge attributes are sharded
        along the edge list, the edge index stays whole, ``pre_process`` shards / embeds the nodes, the block exchanges
        heads around its conv, ``post_process`` extracts (and, for the backward mapper, gathers) the destination rows."""
        from ..distributed.graph import shard_tensor
        from ..distributed.shapes import get_shape_shards

        size = (sum(s_[0] for s_ in shard_shapes[0]), sum(s_[0] for s_ in shard_shapes[1]))
        edge_attr = self.trainable(self.edge_attr, batch_size)
        edge_index = self._expand_edges(self.edge_index_base, self.edge_inc, batch_size)
        shapes_edge_attr = get_shape_shards(edge_attr, 0, model_comm_group)
        edge_attr = shard_tensor(edge_attr, 0, shapes_edge_attr, model_comm_group)
        x_src, x_dst, shapes_src, shapes_dst = self.pre_process(x, shard_shapes, model_comm_group)
        (x_src, x_dst), edge_attr = self.proc((x_src, x_dst), edge_attr, edge_index, (shapes_src, shapes_dst, shapes_edge_attr),
                                              batch_size, model_comm_group, size=size)
        return self.post_process(x_dst, shapes_dst, model_comm_group)

    def _run(self, x, batch_size: int, shard_shapes, model_comm_group) -> Tensor:
        if model_comm_group is not None and model_comm_group.size() > 1:
            assert batch_size == 1, "Only batch size of 1 is supported when model is sharded across GPUs"
            return self._run_sharded(x, batch_size, shard_shapes, model_comm_group)
        x_src, x_dst = x
        if shard_shapes is not None:
            size = (sum(s[0] for s in shard_shapes[0]), sum(s[0] for s in shard_shapes[1]))
            if size != (x_src.shape[0], x_dst.shape[0]):
                raise ValueError(f"shard_shapes describe {size} nodes, inputs have {(x_src.shape[0], x_dst.shape[0])}")
        if training.wants_grad(self, x_src, x_dst):
            with self._offloaded():
                return training.gt_mapper(self, x_src, x_dst, batch_size)
        dtype = runtime.compute_dtype(x_dst)

        def prep(t):
            t = t if t.dtype == dtype else t.to(dtype)
            return t if t.stride(-1) == 1 else t.contiguous()

        return self.native(prep(x_src), prep(x_dst), batch_size)

    def forward(self, x, batch_size: int, shard_shapes, model_comm_group=None) -> Tensor:
        """Reference layers/mapper.py:239-272: pre-process, one mapper block over the sub-graph, post-process -> the
        destination nodes."""
        return self._run(x, batch_size, shard_shapes, model_comm_group)


class GraphTransformerForwardMapper(ForwardMapperPreProcessMixin, GraphTransformerBaseMapper):
    """data -> hidden (reference layers/mapper.py:275-345): both node sets are embedded."""

    def __init__(self, in_channels_src: int = 0, in_channels_dst: int = 0, hidden_dim: int = 128,
                 trainable_size: int = 8, out_channels_dst: Optional[int] = None, num_chunks: int = 1,
                 cpu_offload: bool = False, activation: str = "GELU", num_heads: int = 16, mlp_hidden_ratio: int = 4,
                 sub_graph=None, sub_graph_edge_attributes: Optional[list] = None, src_grid_size: int = 0,
                 dst_grid_size: int = 0) -> None:
        super().__init__(in_channels_src, in_channels_dst, hidden_dim, trainable_size,
                         out_channels_dst=out_channels_dst, num_chunks=num_chunks, cpu_offload=cpu_offload,
                         activation=activation, num_heads=num_heads, mlp_hidden_ratio=mlp_hidden_ratio,
                         sub_graph=sub_graph, sub_graph_edge_attributes=sub_graph_edge_attributes,
                         src_grid_size=src_grid_size, dst_grid_size=dst_grid_size)
        self.emb_nodes_src = nn.Linear(self.in_channels_src, self.hidden_dim)

    def _embed(self, x_src: Tensor, x_dst: Tensor, one_cols=(None, None)):
        eps1, eps2 = self._block_ln_eps(x_src.dtype)  # the embeddings enter the block's layer_norm1 / layer_norm2
        # source rows feed k | v only: their embedding is never written out when it can be folded away
        return (self._embedded("emb_nodes_src", self.emb_nodes_src, x_src, eps1, one_cols[0],
                               materialise=self.proc.update_src_nodes),
                self._embedded("emb_nodes_dst", self.emb_nodes_dst, x_dst, eps2, one_cols[1], materialise=True))

    def forward(self, x, batch_size: int, shard_shapes, model_comm_group=None):
        x_dst = self._run(x, batch_size, shard_shapes, model_comm_group)
        return x[0], x_dst  # the RAW source tensor is handed back (reference layers/mapper.py:344-345)


class GraphTransformerBackwardMapper(BackwardMapperPostProcessMixin, GraphTransformerBaseMapper):
    """hidden -> data (reference layers/mapper.py:348-418): only the destination is embedded, then extracted."""

    def __init__(self, in_channels_src: int = 0, in_channels_dst: int = 0, hidden_dim: int = 128,
                 trainable_size: int = 8, out_channels_dst: Optional[int] = None, num_chunks: int = 1,
                 cpu_offload: bool = False, activation: str = "GELU", num_heads: int = 16, mlp_hidden_ratio: int = 4,
                 sub_graph=None, sub_graph_edge_attributes: Optional[list] = None, src_grid_size: int = 0,
                 dst_grid_size: int = 0) -> None:
        super().__init__(in_channels_src, in_channels_dst, hidden_dim, trainable_size,
                         out_channels_dst=out_channels_dst, num_chunks=num_chunks, cpu_offload=cpu_offload,
                         activation=activation, num_heads=num_heads, mlp_hidden_ratio=mlp_hidden_ratio,
                         sub_graph=sub_graph, sub_graph_edge_attributes=sub_graph_edge_attributes,
                         src_grid_size=src_grid_size, dst_grid_size=dst_grid_size)
        self.node_data_extractor = nn.Sequential(nn.LayerNorm(self.hidden_dim),
                                                 nn.Linear(self.hidden_dim, self.out_channels_dst))

    def pre_process(self, x, shard_shapes, model_comm_group=None):
        """Only the destination is embedded; the source already lives in the hidden space (reference :412-418)."""
        from ..distributed.shapes import change_channels_in_shape

        from ..distributed.graph import shard_tensor

        x_src, x_dst, shapes_src, shapes_dst = super().pre_process(x, shard_shapes, model_comm_group)
        x_dst = shard_tensor(x_dst, 0, shapes_dst, model_comm_group)  # (identity without a model group)
        return (x_src, self._apply_module(self.emb_nodes_dst, x_dst), change_channels_in_shape(shapes_src, self.hidden_dim),
                change_channels_in_shape(shapes_dst, self.hidden_dim))

    def _embed(self, x_src: Tensor, x_dst: Tensor, one_cols=(None, None)):
        return x_src, self._embedded("emb_nodes_dst", self.emb_nodes_dst, x_dst, self._block_ln_eps(x_dst.dtype)[1],
                                     one_cols[1], materialise=True)

    def _extract(self, x_dst: Tensor, out_dtype) -> Tensor:
        ln, lin = self.node_data_extractor[0], self.node_data_extractor[1]
        if runtime.ln_fold_enabled(x_dst.dtype) and x_dst.shape[1] % ops.k_multiple(x_dst.dtype) == 0:
            # LayerNorm folded into the extraction Linear (row statistics + anemoi_linear_ln)
            wf, bf, cs = self._packed.get(("extract", "lnfold", x_dst.dtype), [lin.weight, lin.bias, ln.weight, ln.bias],
                                          lambda: runtime.fold_layer_norm(lin.weight.detach().float(), lin.bias,
                                                                          ln.weight, ln.bias, x_dst.dtype))
            return ops.linear(x_dst, wf, bf, out_dtype=out_dtype, ln=(ops.row_stats(x_dst, ln.eps), cs))
        h = ops.layer_norm(x_dst, runtime.f32c(ln.weight), runtime.f32c(ln.bias), ln.eps)
        return linear_native(self._packed, "extract", lin, h, out_dtype=out_dtype)

    def forward(self, x, batch_size: int, shard_shapes, model_comm_group=None) -> Tensor:
        return self._run(x, batch_size, shard_shapes, model_comm_group)


# ---------------------------------------------------------------------------------------------
# GNN mappers: parameter layout mirrors reference layers/mapper.py:421-705
# ---------------------------------------------------------------------------------------------
class GNNBaseMapper(GraphEdgeMixin, BaseMapper):
    def __init__(self, in_channels_src: int = 0, in_channels_dst: int = 0, hidden_dim: int = 128,
                 trainable_size: int = 8, out_channels_dst: Optional[int] = None, num_chunks: int = 1,
                 cpu_offload: bool = False, activation: str = "SiLU", mlp_extra_layers: int = 0, sub_graph=None,
                 sub_graph_edge_attributes: Optional[list] = None, src_grid_size: int = 0,
                 dst_grid_size: int = 0) -> None:
        super().__init__(in_channels_src, in_channels_dst, hidden_dim, out_channels_dst=out_channels_dst,
                         num_chunks=num_chunks, cpu_offload=cpu_offload, activation=activation)
        self._register_edges(sub_graph, sub_graph_edge_attributes, src_grid_size, dst_grid_size, trainable_size)
        self.emb_edges = MLP(in_features=self.edge_dim, hidden_dim=hidden_dim, out_features=hidden_dim,
                             n_extra_layers=mlp_extra_layers, activation=activation)
        self.trainable = TrainableTensor(trainable_size=trainable_size, tensor_size=self.edge_attr.shape[0])
        self._plans = runtime.PlanCache()

    def prepare_edges(self, size, batch_size: int, model_comm_group=None):
        """Reference layers/mapper.py:485-495: ``(emb_edges(cat[edge_attr, trainable]) [E * batch, hidden], edge_index
        [2, E * batch])`` in the sub-graph's edge order (a single model rank keeps that order; the reference only
        re-sorts for its 1-hop sharding).  The mapper forward does not come through here: it embeds the edges already
        in destination-sorted order."""
        from ..distributed.graph import shard_tensor
        from ..distributed.khop_edges import sort_edges_1hop_sharding

        edge_attr = self.trainable(self.edge_attr, batch_size)
        edge_index = self._expand_edges(self.edge_index_base, self.edge_inc, batch_size)
        if model_comm_group is not None and model_comm_group.size() > 1:  # this rank's 1-hop edge shard
            edge_attr, edge_index, shapes_edge_attr, shapes_edge_idx = sort_edges_1hop_sharding(size, edge_attr, edge_index,
                                                                                                model_comm_group)
            edge_index = shard_tensor(edge_index, 1, shapes_edge_idx, model_comm_group)
            edge_attr = shard_tensor(edge_attr, 0, shapes_edge_attr, model_comm_group)
        return self.emb_edges(edge_attr), edge_index

    # hooks: the forward mapper embeds both node sets, the backward mapper extracts the output variables
    def _embed(self, x_src: Tensor, x_dst: Tensor):
        return x_src, x_dst

    def _extract(self, x_dst: Tensor, out_dtype) -> Tensor:
        return x_dst

    def native(self, x_src: Tensor, x_dst: Tensor, batch_size: int, out_dtype: Optional[torch.dtype] = None,
               src_map: Optional[Tensor] = None, dst_map: Optional[Tensor] = None):
        """Reference layers/mapper.py:485-522: embed edges, embed nodes, one GraphConvMapperBlock, post-process."""
        n_src, n_dst = x_src.shape[0], x_dst.shape[0]
        plan = self._plans.get(self.edge_index_base, n_src, n_dst, batch_size, self.edge_inc, src_map, dst_map)
        dtype = x_dst.dtype
        ea = ops.edge_attr_csr(self.edge_attr, self.trainable.trainable, plan.perm)
        e = ops.convert_pad(ea[:, : self.edge_dim], dtype, ops.round_up(self.edge_dim, ops.k_multiple(dtype)))
        e = self.emb_edges.native()(e)
        h_src, h_dst = self._embed(x_src, x_dst)
        (h_src, h_dst), _ = self.proc.native(h_src, h_dst, e, plan)
        return h_src, self._extract(h_dst, out_dtype)

    def native_local(self, x_src: Tensor, x_dst: Tensor, local_graph, out_dtype: Optional[torch.dtype] = None,
                     x_src_extra: Optional[Tensor] = None):
        """Node-partitioned run (``distributed/partition.py``): local source / destination rows, local CSR plan (``perm``
        = original edge ids), halo source rows (backward mapper) by all-to-all-v inside the block.  ``x_src_extra``
        (forward mapper): further SOURCE rows whose updated embedding the caller needs (the grid rows this rank decodes):
        the source update of the block is row-local (reference layers/block.py:282), so it is evaluated on them too.
        Returns ``(updated source rows of x_src_extra or None, destination rows)``."""
        plan = local_graph.plan
        dtype = x_dst.dtype
        ea = ops.edge_attr_csr(self.edge_attr, self.trainable.trainable, plan.perm)
        e = ops.convert_pad(ea[:, : self.edge_dim], dtype, ops.round_up(self.edge_dim, ops.k_multiple(dtype)))
        e = self.emb_edges.native()(e)
        h_src, h_dst = self._embed(x_src, x_dst)
        (_, h_dst), _ = self.proc.native(h_src, h_dst, e, plan, local_graph.halo)
        extra = None
        if x_src_extra is not None:
            hx = self.emb_nodes_src.native()(x_src_extra) if hasattr(self, "emb_nodes_src") else x_src_extra
            c = hx.shape[1]
            if self.proc.update_src_nodes:
                xcat = torch.empty((hx.shape[0], 2 * c), dtype=dtype, device=hx.device)
                xcat[:, :c].copy_(hx)
                xcat[:, c:].copy_(hx)
                hx = self.proc.node_mlp.native()(xcat, residual=hx)
            extra = hx
        return extra, self._extract(h_dst, out_dtype)

    def _run(self, x, batch_size: int, shard_shapes, model_comm_group):
        if model_comm_group is not None and model_comm_group.size() > 1:
            # the reference's mapper forward across a model group (layers/mapper.py:497-522)
            assert batch_size == 1, "Only batch size of 1 is supported when model is sharded across GPUs"
            size = (sum(s_[0] for s_ in shard_shapes[0]), sum(s_[0] for s_ in shard_shapes[1]))
            edge_attr, edge_index = self.prepare_edges(size, batch_size, model_comm_group)
            x_src, x_dst, shapes_src, shapes_dst = self.pre_process(x, shard_shapes, model_comm_group)
            (x_src, x_dst), _ = self.proc((x_src, x_dst), edge_attr, edge_index, (shapes_src, shapes_dst), model_comm_group,
                                          size=size)
            return x_src, self.post_process(x_dst, shapes_dst, model_comm_group)
        x_src, x_dst = x
        if training.wants_grad(self, x_src, x_dst):
            with self._offloaded():
                return training.gnn_mapper(self, x_src, x_dst, batch_size)
        dtype = runtime.compute_dtype(x_dst)

        def prep(t):
            t = t if t.dtype == dtype else t.to(dtype)
            return t if t.stride(-1) == 1 else t.contiguous()

        return self.native(prep(x_src), prep(x_dst), batch_size)

    def forward(self, x, batch_size: int, shard_shapes, model_comm_group=None):
        return self._run(x, batch_size, shard_shapes, model_comm_group)


class GNNForwardMapper(ForwardMapperPreProcessMixin, GNNBaseMapper):
    def __init__(self, in_channels_src: int = 0, in_channels_dst: int = 0, hidden_dim: int = 128,
                 trainable_size: int = 8, out_channels_dst: Optional[int] = None, num_chunks: int = 1,
                 cpu_offload: bool = False, activation: str = "SiLU", mlp_extra_layers: int = 0, sub_graph=None,
                 sub_graph_edge_attributes: Optional[list] = None, src_grid_size: int = 0,
                 dst_grid_size: int = 0) -> None:
        super().__init__(in_channels_src, in_channels_dst, hidden_dim, trainable_size, out_channels_dst, num_chunks,
                         cpu_offload, activation, mlp_extra_layers, sub_graph=sub_graph,
                         sub_graph_edge_attributes=sub_graph_edge_attributes, src_grid_size=src_grid_size,
                         dst_grid_size=dst_grid_size)
        self.proc = GraphConvMapperBlock(hidden_dim, hidden_dim, mlp_extra_layers=mlp_extra_layers,
                                         activation=activation, update_src_nodes=True, num_chunks=num_chunks)
        self.offload_layers(cpu_offload)
        self.emb_nodes_src = MLP(in_features=in_channels_src, hidden_dim=hidden_dim, out_features=hidden_dim,
                                 n_extra_layers=mlp_extra_layers, activation=activation)
        self.emb_nodes_dst = MLP(in_features=in_channels_dst, hidden_dim=hidden_dim, out_features=hidden_dim,
                                 n_extra_layers=mlp_extra_layers, activation=activation)

    def _embed(self, x_src: Tensor, x_dst: Tensor):
        return self.emb_nodes_src.native()(x_src), self.emb_nodes_dst.native()(x_dst)


class GNNBackwardMapper(BackwardMapperPostProcessMixin, GNNBaseMapper):
    def __init__(self, in_channels_src: int = 0, in_channels_dst: int = 0, hidden_dim: int = 128,
                 trainable_size: int = 8, out_channels_dst: Optional[int] = None, num_chunks: int = 1,
                 cpu_offload: bool = False, activation: str = "SiLU", mlp_extra_layers: int = 0, sub_graph=None,
                 sub_graph_edge_attributes: Optional[list] = None, src_grid_size: int = 0,
                 dst_grid_size: int = 0) -> None:
        super().__init__(in_channels_src, in_channels_dst, hidden_dim, trainable_size,
                         out_channels_dst=out_channels_dst, num_chunks=num_chunks, cpu_offload=cpu_offload,
                         activation=activation, mlp_extra_layers=mlp_extra_layers, sub_graph=sub_graph,
                         sub_graph_edge_attributes=sub_graph_edge_attributes, src_grid_size=src_grid_size,
                         dst_grid_size=dst_grid_size)
        self.proc = GraphConvMapperBlock(hidden_dim, hidden_dim, mlp_extra_layers=mlp_extra_layers,
                                         activation=activation, update_src_nodes=False, num_chunks=num_chunks)
        self.offload_layers(cpu_offload)
        self.node_data_extractor = MLP(in_features=self.hidden_dim, hidden_dim=self.hidden_dim,
                                       out_features=self.out_channels_dst, n_extra_layers=mlp_extra_layers,
                                       activation=self.activation, layer_norm=False, final_activation=False)

    def pre_process(self, x, shard_shapes, model_comm_group=None):
        """Both node sets already live in the hidden space: only the shapes change (reference :690-694)."""
        from ..distributed.shapes import change_channels_in_shape

        x_src, x_dst, shapes_src, shapes_dst = super().pre_process(x, shard_shapes, model_comm_group)
        return (x_src, x_dst, change_channels_in_shape(shapes_src, self.hidden_dim),
                change_channels_in_shape(shapes_dst, self.hidden_dim))

    def _extract(self, x_dst: Tensor, out_dtype) -> Tensor:
        return self.node_data_extractor.native()(x_dst, out_dtype=out_dtype)

    def forward(self, x, batch_size: int, shard_shapes, model_comm_group=None) -> Tensor:
        return self._run(x, batch_size, shard_shapes, model_comm_group)[1]
