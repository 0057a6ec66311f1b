"""Processors mirroring reference layers/processor.py: a stack of ``num_layers`` blocks in ``num_chunks`` chunks.

A processor owns the mesh edge set (non-persistent buffers ``edge_attr`` / ``edge_index_base``, persistent
``edge_inc``) and the per-edge trainable tensor; once per forward it builds / reuses the destination-sorted CSR
plan and gathers ``[edge_attr | trainable]`` into CSR order, then every block runs on that shared edge input.
"""

from __future__ import annotations

import os

from abc import ABC
from typing import Optional

import torch
from torch import Tensor
from torch import nn

from .. import ops
from .. import runtime
from .. import training
from .chunk import GNNProcessorChunk
from .chunk import GraphTransformerProcessorChunk
from .chunk import TransformerProcessorChunk
from .graph import TrainableTensor
from .mapper import GraphEdgeMixin


class BaseProcessor(nn.Module, ABC):
    def __init__(self, num_layers: int, *args, num_channels: int = 128, num_chunks: int = 2,
                 activation: str = "GELU", cpu_offload: bool = False, **kwargs) -> None:
        super().__init__()
        self.num_chunks = num_chunks
        self.num_channels = num_channels
        self.chunk_size = num_layers // num_chunks
        assert (
            num_layers % num_chunks == 0
        ), f"Number of processor layers ({num_layers}) has to be divisible by the number of processor chunks ({num_chunks})."

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

    def build_layers(self, processor_chunk_class, *args, **kwargs) -> None:
        self.proc = nn.ModuleList([processor_chunk_class(*args, **kwargs) for _ in range(self.num_chunks)])

    def run_layers(self, data: tuple, *args, **kwargs):
        for layer in self.proc:
            data = layer(*data, *args, **kwargs)
        return data

    def forward(self, x: Tensor, *args, **kwargs) -> Tensor:
        return self.run_layers((x,), *args, **kwargs)


class TransformerProcessor(BaseProcessor):
    def __init__(self, num_layers: int, *args, window_size: Optional[int] = None, num_channels: int = 128,
                 num_chunks: int = 2, activation: str = "GELU", cpu_offload: bool = False, num_heads: int = 16,
                 mlp_hidden_ratio: int = 4, dropout_p: float = 0.1, **kwargs) -> None:
        super().__init__(num_channels=num_channels, num_layers=num_layers, num_chunks=num_chunks,
                         activation=activation, cpu_offload=cpu_offload)
        self.build_layers(
            TransformerProcessorChunk, num_channels=num_channels, mlp_hidden_ratio=mlp_hidden_ratio,
            num_heads=num_heads, num_layers=self.chunk_size, window_size=window_size, activation=activation,
            dropout_p=dropout_p,
        )
        self.offload_layers(cpu_offload)

    def native(self, x: Tensor, batch_size: int, node_map: Optional[Tensor] = None) -> Tensor:
        """x ``[B * N, C]`` in the compute dtype.  Global attention is permutation-equivariant, so an internal node
        order needs no special handling; a sliding window acts on the EXTERNAL order, hence the un/re-permutation."""
        windowed = self.proc[0].blocks[0].attention.attention_window() >= 0
        if node_map is not None and windowed:
            n = node_map.shape[0]
            ext = torch.cat([node_map + b * n for b in range(batch_size)])  # external row -> internal row
            x = x.index_select(0, ext)
        for chunk in self.proc:
            x = chunk.native(x, batch_size)
        if node_map is not None and windowed:
            back = torch.empty_like(x)
            back[ext] = x
            x = back
        return x

    def native_local(self, x_own: Tensor, local_graph) -> Tensor:
        """Node-partitioned run: rows stay sharded for LayerNorm / Linear / MLP; around the attention the heads <-> rows
        all-to-all of the reference (distributed/transformer.py:85-130) gives every rank all rows of H / P heads."""
        for chunk in self.proc:
            for blk in chunk.blocks:
                x_own = blk.native(x_own, 1, head_exchange=local_graph.heads)
        return x_own

    def forward(self, x: Tensor, batch_size: int, shard_shapes, model_comm_group=None, *args, **kwargs) -> Tensor:
        if model_comm_group is not None:
            assert (
                model_comm_group.size() == 1 or batch_size == 1
            ), "Only batch size of 1 is supported when model is sharded accross GPUs"
            if model_comm_group.size() > 1:
                # the reference's protocol (layers/processor.py:103-137): x is this rank's row shard, shard_shapes the row
                # counts of all ranks; every block reshards rows <-> heads around its attention
                for chunk in self.proc:
                    for blk in chunk.blocks:
                        x = blk(x, shard_shapes, batch_size, model_comm_group)
                return x
        if training.wants_grad(self, x):
            with self._offloaded():
                return training.transformer_processor(self, x, batch_size)
        dtype = runtime.compute_dtype(x)
        xin = x if x.dtype == dtype else x.to(dtype)
        return self.native(xin if xin.stride(-1) == 1 else xin.contiguous(), batch_size)


class GNNProcessor(GraphEdgeMixin, BaseProcessor):
    def __init__(self, num_layers: int, *args, trainable_size: int = 8, num_channels: int = 128, num_chunks: int = 2,
                 mlp_extra_layers: int = 0, activation: str = "SiLU", cpu_offload: bool = False, sub_graph=None,
                 sub_graph_edge_attributes: Optional[list] = None, src_grid_size: int = 0, dst_grid_size: int = 0,
                 **kwargs) -> None:
        super().__init__(num_channels=num_channels, num_layers=num_layers, num_chunks=num_chunks,
                         activation=activation, cpu_offload=cpu_offload)
        self._register_edges(sub_graph, sub_graph_edge_attributes, src_grid_size, dst_grid_size, trainable_size)
        self.trainable = TrainableTensor(trainable_size=trainable_size, tensor_size=self.edge_attr.shape[0])
        kw = {"num_layers": self.chunk_size, "mlp_extra_layers": mlp_extra_layers, "activation": activation,
              "edge_dim": None}
        self.build_layers(GNNProcessorChunk, num_channels, **kw)
        kw["edge_dim"] = self.edge_dim  # only the first chunk embeds the raw edge attributes
        self.proc[0] = GNNProcessorChunk(num_channels, **kw)
        self.offload_layers(cpu_offload)

        self._plans = runtime.PlanCache()

    def native(self, x: Tensor, batch_size: int, node_map: Optional[Tensor] = None) -> Tensor:
        """x ``[B * N, C]`` in the compute dtype -> processed nodes.  The edge state lives in CSR order throughout."""
        n = x.shape[0]
        plan = self._plans.get(self.edge_index_base, n, n, batch_size, self.edge_inc, node_map, node_map)
        ea = ops.edge_attr_csr(self.edge_attr, self.trainable.trainable, plan.perm)  # [E, pad4(edge_dim)] f32
        e = ops.convert_pad(ea[:, : self.edge_dim], x.dtype, ops.round_up(self.edge_dim, ops.k_multiple(x.dtype)))
        for chunk in self.proc:
            x, e = chunk.native(x, e, plan)
        return x

    def native_local(self, x_own: Tensor, local_graph) -> Tensor:
        """Node-partitioned run: this rank's mesh rows in / out; the edge state of the edges it owns stays local."""
        plan = local_graph.plan
        ea = ops.edge_attr_csr(self.edge_attr, self.trainable.trainable, plan.perm)
        e = ops.convert_pad(ea[:, : self.edge_dim], x_own.dtype,
                            ops.round_up(self.edge_dim, ops.k_multiple(x_own.dtype)))
        for chunk in self.proc:
            x_own, e = chunk.native(x_own, e, plan, local_graph.halo)
        return x_own

    def forward(self, x: Tensor, batch_size: int, shard_shapes, model_comm_group=None) -> Tensor:
        if model_comm_group is not None and model_comm_group.size() > 1:
            # the reference's protocol (layers/processor.py:228-250): edges sorted into the 1-hop neighbourhoods of the
            # ranks' destination ranges, attributes AND index sharded; x is this rank's row shard
            from ..distributed.graph import shard_tensor
            from ..distributed.khop_edges import sort_edges_1hop_sharding
            from ..distributed.shapes import change_channels_in_shape

            shape_nodes = change_channels_in_shape(shard_shapes, self.num_channels)
            edge_attr = self.trainable(self.edge_attr, batch_size)
            edge_index = self._expand_edges(self.edge_index_base, self.edge_inc, batch_size)
            target_nodes = sum(s_[0] for s_ in shape_nodes)
            edge_attr, edge_index, shapes_edge_attr, shapes_edge_idx = sort_edges_1hop_sharding(
                target_nodes, edge_attr, edge_index, model_comm_group)
            edge_index = shard_tensor(edge_index, 1, shapes_edge_idx, model_comm_group)
            edge_attr = shard_tensor(edge_attr, 0, shapes_edge_attr, model_comm_group)
            for chunk in self.proc:
                x, edge_attr = chunk(x, edge_attr, edge_index, (shape_nodes, shape_nodes), model_comm_group,
                                     size=(target_nodes, target_nodes))
            return x
        if training.wants_grad(self, x):
            with self._offloaded():
                return training.gnn_processor(self, x, batch_size)
        dtype = runtime.compute_dtype(x)
        xin = x if x.dtype == dtype else x.to(dtype)
        return self.native(xin if xin.stride(-1) == 1 else xin.contiguous(), batch_size)


class _BlockAbiPlan:
    """Argument-block TEMPLATES (``anemoi_gt_block_args``) of every block of a GraphTransformer processor for one (row
    count, device, weights version): the packed weights' pointers, shapes and epsilons are filled in once; a forward copies
    the templates, points them at intermediates it allocates itself (``torch.empty``: stream-ordered, legal under stream
    capture) and makes ``num_layers`` FFI calls.  Nothing a launch writes to is owned by the plan, so a replaced plan
    (another row count, new weights) cannot free memory a captured HIP graph still addresses -- a graph's intermediates
    live in ITS memory pool --, and ``run`` is re-entrant across streams.  The node matrix ping-pongs between two buffers
    of the call, the last block writes the result tensor."""

    def __init__(self) -> None:
        self.ok = False
        self.sig = None

    @classmethod
    def build(cls, proc, x: Tensor, ea: Tensor, plan, sig) -> "_BlockAbiPlan":
        import ctypes

        from .. import _lib

        self = cls()
        self.sig = sig
        blocks = [blk for chunk in proc.proc for blk in chunk.blocks]
        dtype, n = x.dtype, x.shape[0]
        operands = []
        for blk in blocks:
            all4 = [blk.lin_self, blk.lin_query, blk.lin_key, blk.lin_value]
            op = blk.block_abi_operands(dtype, all4, "sqkvu")
            if op is None:
                return self
            operands.append(op)
        blk0 = blocks[0]
        c, h = blk0.num_heads * blk0.out_channels_conv, blk0.num_heads
        up = operands[0]["up"]
        n_in, k_proj, hidden = operands[0]["w_in"].shape[0], operands[0]["w_proj"].shape[1], operands[0]["w_fc1"].shape[0]
        if x.shape[1] != c or any(o["w_in"].shape != operands[0]["w_in"].shape or o["w_fc1"].shape[0] != hidden
                                  or o["act"] not in _lib.ACT_CODES for o in operands):
            return self
        from .block import edge_schedule, edge_tiles, set_tile_args

        tiles = edge_tiles(plan, x, h, up)
        sched = None if tiles is not None else edge_schedule(plan, x)  # (n_edges below: the entry point declines it beyond 32-bit attribute-row offsets)
        self.keep = [operands, ea, plan, sched, tiles]  # the packed weights, edge attributes, CSR and schedule the templates point at
        self.dims = (n, c, h, up, n_in, k_proj, hidden)
        lib = _lib.load()
        self.ws_bytes = n * max(c // 128, 1) * 8  # row-sum partials of anemoi_linear_stats
        self.eps_in = operands[0]["eps_ln1"]
        self.args = []
        for i, (blk, o) in enumerate(zip(blocks, operands)):
            a = _lib.GtBlockArgs()
            a.struct_bytes = ctypes.sizeof(_lib.GtBlockArgs)
            a.n_dst, a.dtype, a.C, a.H, a.up = n, ops.dtype_code(dtype), c, h, up
            a.hidden, a.act, a.k_proj, a.n_in = hidden, _lib.ACT_CODES[o["act"]], k_proj, n_in
            a.eps_mlp = o["eps_mlp"]
            nxt = blocks[i + 1] if i + 1 < len(blocks) else blk  # the LayerNorm that reads this block's output next
            a.eps_out = nxt.layer_norm1.eps
            a.ldx = c
            a.w_in, a.cs_in = o["w_in"].data_ptr(), o["cs_in"].data_ptr()
            a.b_in = None if o["b_in"] is None else o["b_in"].data_ptr()
            a.ld_sq = n_in
            a.edge_attr, a.rowptr, a.col = ea.data_ptr(), plan.rowptr.data_ptr(), plan.col.data_ptr()
            if tiles is not None:
                set_tile_args(a, tiles, plan.n_src, ea.shape[0])
            if sched is not None:
                a.sched, a.sched_slots, a.sched_steps, a.n_src = sched.data_ptr(), sched.shape[1], sched.shape[2], plan.n_src
                a.n_edges = ea.shape[0]
            a.ld_att = k_proj
            a.w_proj = o["w_proj"].data_ptr()
            a.b_proj = None if o["b_proj"] is None else o["b_proj"].data_ptr()
            a.w_fc1, a.cs_fc1 = o["w_fc1"].data_ptr(), o["cs_fc1"].data_ptr()
            a.b_fc1 = None if o["b_fc1"] is None else o["b_fc1"].data_ptr()
            a.w_fc2 = o["w_fc2"].data_ptr()
            a.b_fc2 = None if o["b_fc2"] is None else o["b_fc2"].data_ptr()
            self.args.append(a)
        self.fn = lib.anemoi_gt_processor_block_forward
        self.eps_out = blocks[-1].layer_norm1.eps
        self.ok = True
        return self

    def run(self, x: Tensor) -> Tensor:
        import ctypes

        from .. import _lib

        n, c, h, up, n_in, k_proj, hidden = self.dims
        dtype, dev = x.dtype, x.device
        new = lambda *shape, dt=dtype: torch.empty(shape, dtype=dt, device=dev)  # noqa: E731
        stats_in = ops.row_stats(x, self.eps_in)  # carried by the GEMM that produced x, or one pass over it
        # this call's intermediates (shared by its blocks: they run back to back on one stream)
        sq, att, y, hbuf = new(n, n_in), new(n, k_proj), new(n, c), new(n, hidden)
        if k_proj > c + h * up:
            att[:, c + h * up:].zero_()  # the K padding of the projection: the edge kernel never writes it
        bufs = [new(n, c), new(n, c)]
        stats = [new(n, 2, dt=torch.float32) for _ in range(3)]  # y, out (ping), out (pong)
        ws = new(self.ws_bytes, dt=torch.uint8)
        out, out_stats = new(n, c), new(n, 2, dt=torch.float32)
        stream = ops._stream()
        last = len(self.args) - 1
        struct = _lib.GtBlockArgs
        for i, template in enumerate(self.args):
            a = struct.from_buffer_copy(template)
            a.sq, a.att, a.y, a.y_stats, a.h = sq.data_ptr(), att.data_ptr(), y.data_ptr(), stats[0].data_ptr(), hbuf.data_ptr()
            a.stats_ws, a.stats_ws_bytes = ws.data_ptr(), self.ws_bytes
            # block i reads buffer (i - 1) % 2 (block 0: the caller's x) and writes buffer i % 2 (last block: the result)
            if i == 0:
                a.x, a.x_stats = x.data_ptr(), stats_in.data_ptr()
            else:
                a.x, a.x_stats = bufs[(i - 1) % 2].data_ptr(), stats[1 + (i - 1) % 2].data_ptr()
            if i == last:
                a.out, a.out_stats = out.data_ptr(), out_stats.data_ptr()
            else:
                a.out, a.out_stats = bufs[i % 2].data_ptr(), stats[1 + i % 2].data_ptr()
            st = self.fn(ctypes.byref(a), stream)
            if st != 0:
                _lib.check(st, "anemoi_gt_processor_block_forward")
        ops._carry_stats(out, self.eps_out, out_stats)
        return out


class GraphTransformerProcessor(GraphEdgeMixin, BaseProcessor):
    def __init__(self, num_layers: int, trainable_size: int = 8, num_channels: int = 128, num_chunks: int = 2,
                 num_heads: int = 16, mlp_hidden_ratio: int = 4, activation: str = "GELU", cpu_offload: bool = False,
                 sub_graph=None, sub_graph_edge_attributes: Optional[list] = None, src_grid_size: int = 0,
                 dst_grid_size: int = 0, **kwargs) -> None:
        super().__init__(num_layers=num_layers, num_channels=num_channels, num_chunks=num_chunks,
                         activation=activation, cpu_offload=cpu_offload)
        self._register_edges(sub_graph, sub_graph_edge_attributes, src_grid_size, dst_grid_size, trainable_size)
        self.trainable = TrainableTensor(trainable_size=trainable_size, tensor_size=self.edge_attr.shape[0])
        self.build_layers(
            GraphTransformerProcessorChunk, num_channels=num_channels, num_layers=self.chunk_size,
            num_heads=num_heads, mlp_hidden_ratio=mlp_hidden_ratio, activation=activation, edge_dim=self.edge_dim,
        )
        self.offload_layers(cpu_offload)
        self._plans = runtime.PlanCache()
        self._packed = runtime.PackedWeights()

    def native(self, x: Tensor, batch_size: int, node_map: Optional[Tensor] = None) -> Tensor:
        """x ``[B * N, C]`` in the compute dtype -> processed nodes (same dtype).

        ``node_map``: optional external-id -> row relabelling when ``x`` is kept in an internal node order."""
        n = x.shape[0]
        plan = self._plans.get(self.edge_index_base, n, n, batch_size, self.edge_inc, node_map, node_map)
        ea = runtime.edge_attr_csr_cached(self._packed, self.edge_attr, self.trainable.trainable, plan,
                                          *self.proc[0].blocks[0].edge_layout(x.dtype))
        fast = self._block_abi_plan(x, ea, plan)
        if fast is not None:
            return fast.run(x)
        for chunk in self.proc:
            x = chunk.native(x, ea, plan)
        return x

    # ---- block-level C ABI: one FFI call per block (anemoi_gt_processor_block_forward), same kernels, same packed weights
    block_abi = True  # False: always the op-by-op route (tests compare the two; bench.py's instrumented pass times ops)

    def _block_abi_plan(self, x: Tensor, ea: Tensor, plan):
        if not self.block_abi or ops.PROFILE is not None or not x.is_cuda or x.dtype != torch.bfloat16 or x.shape[0] == 0:
            return None
        if x.stride(1) != 1 or x.stride(0) != x.shape[1] or plan.num_edges == 0:
            return None
        params = self.__dict__.get("_abi_params")
        if params is None:
            params = self.__dict__["_abi_params"] = [p for p in self.parameters()]
        # any in-place change of a parameter (optimiser step, load_state_dict, .normal_()) bumps its version counter
        sig = (sum(p._version for p in params), params[0].data_ptr(), x.shape[0], str(x.device), ea.data_ptr(), id(plan),
               os.environ.get("ANEMOI_AMD_EDGE_TILES"), os.environ.get("ANEMOI_AMD_EDGE_SCHED"),  # (A/B switches of the edge
               os.environ.get("ANEMOI_AMD_LN_FOLD"))  # kernel; the fold the plan's operands are built for: a model that ran
        # with the fold and is then asked without it kept the folded blocks between unfolded mappers until round 6)
        fast = self.__dict__.get("_abi_plan")
        if fast is None or fast.sig != sig:
            fast = self.__dict__["_abi_plan"] = _BlockAbiPlan.build(self, x, ea, plan, sig)
        return fast if fast.ok else None

    def _apply(self, fn, *args, **kwargs):  # .to() / .cuda() / .bfloat16(): new storages behind the cached pointers
        self.__dict__.pop("_abi_plan", None)
        self.__dict__.pop("_abi_params", None)
        return super()._apply(fn, *args, **kwargs)

    def native_local(self, x_own: Tensor, local_graph) -> Tensor:
        """Node-partitioned run (``distributed/partition.py``): this rank's mesh rows in, same rows out."""
        plan = local_graph.plan
        ea = runtime.edge_attr_csr_cached(self._packed, self.edge_attr, self.trainable.trainable, plan,
                                          *self.proc[0].blocks[0].edge_layout(x_own.dtype))
        for chunk in self.proc:
            x_own = chunk.native(x_own, ea, plan, local_graph.halo)
        return x_own

    def forward(self, x: Tensor, batch_size: int, shard_shapes, model_comm_group=None, *args, **kwargs) -> Tensor:
        if model_comm_group is not None and model_comm_group.size() > 1:
            assert batch_size == 1, "Only batch size of 1 is supported when model is sharded across GPUs"
            # the reference's protocol (layers/processor.py:317-343): x is this rank's row shard; the edge attributes are
            # sharded along the edge list, the edge index stays whole; every block exchanges heads around its conv
            from ..distributed.graph import shard_tensor
            from ..distributed.shapes import get_shape_shards

            edge_attr = self.trainable(self.edge_attr, batch_size)
            edge_index = self._expand_edges(self.edge_index_base, self.edge_inc, batch_size)
            shapes_edge_attr = get_shape_shards(edge_attr, 0, model_comm_group)
            edge_attr = shard_tensor(edge_attr, 0, shapes_edge_attr, model_comm_group)
            n_all = sum(s_[0] for s_ in shard_shapes)
            for chunk in self.proc:
                for blk in chunk.blocks:
                    x, edge_attr = blk(x, edge_attr, edge_index, (shard_shapes, shard_shapes, shapes_edge_attr), batch_size,
                                       model_comm_group, size=(n_all, n_all))
            return x
        if training.wants_grad(self, x):
            with self._offloaded():
                return training.gt_processor(self, x, batch_size)
        dtype = runtime.compute_dtype(x)
        xin = x if x.dtype == dtype else x.to(dtype)
        return self.native(xin if xin.stride(-1) == 1 else xin.contiguous(), batch_size)
