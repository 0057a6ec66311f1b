"""Node-partitioned forward of ``AnemoiModelEncProcDec`` over one model communication group.

Partition (built once per (graph, world size, rank) and cached on the model):

* mesh nodes, in the internal Morton order, are cut into ``P`` contiguous ranges (``tensor_split`` sizes); rank ``r``
  owns range ``r`` -- i.e. a spatially compact patch of the sphere;
* processor: rank ``r`` owns the edges whose destination it owns (exactly the dst-partition of the reference's
  ``sort_edges_1hop_chunks``, reference distributed/khop_edges.py:88-130); sources outside the range are its HALO.
  Per block one all-to-all-v moves the k|v rows of halo nodes from their owners (2C values per halo node);
* encoder: every rank receives the full input (contract), so it embeds and projects exactly the grid rows that feed
  its mesh nodes -- no communication at all;
* decoder: grid node ``g`` belongs to the rank that owns its first (nearest) mesh source, so decoder edges are mostly
  local; one all-to-all-v of mesh k|v halo rows; the ``[rows, V_out]`` results are all-gathered (padded) and put back
  in grid order on every rank.

No all-reduce anywhere in the forward.  All index plumbing is torch; the arithmetic goes through ``ops`` as on one GPU.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import torch
import torch.distributed as dist
from torch import Tensor

from .. import ops
from .. import runtime
from ..runtime import EdgePlan
from .khop_edges import edge_chunk_ids, partition_edges_by_dst
from .shapes import split_bounds


class SimulatedRank:
    """Stand-in group for timing ONE rank of a ``world``-way partition alone (tools/sim_rank.py): the plans, shapes and
    kernel launches are those of rank ``rank``; the exchanges move no data (halo rows read as zeros), so the results
    are meaningless and only the compute side of the step is timed.  Never used by the product path."""

    def __init__(self, rank: int, world: int) -> None:
        self.rank_, self.world_ = rank, world

    def size(self) -> int:
        return self.world_

    def rank(self) -> int:
        return self.rank_


def _rank(group) -> int:
    return group.rank_ if isinstance(group, SimulatedRank) else dist.get_rank(group)


def _world(group) -> int:
    return group.world_ if isinstance(group, SimulatedRank) else dist.get_world_size(group)


def _backend(group) -> str:
    return "sim" if isinstance(group, SimulatedRank) else dist.get_backend(group)


def _alltoallv(out: Tensor, inp: Tensor, out_splits: List[int], in_splits: List[int], group, async_op: bool = False):
    """Row-wise all-to-all-v.  RCCL path: one ``all_to_all_single`` (optionally asynchronous: it runs on RCCL's stream
    and the returned work handle is waited for right before the consumer); other backends (gloo in CPU tests, or gloo
    over device tensors when several debugging ranks share one GPU): blocking P2P, staged through host memory."""
    backend = _backend(group)
    if backend == "nccl":
        return dist.all_to_all_single(out, inp, out_splits, in_splits, group=group, async_op=async_op)
    if backend == "sim":
        out.zero_()
        return None
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    staged = inp.is_cuda
    h_out = torch.empty(out.shape, dtype=out.dtype) if staged else out
    h_inp = inp.cpu() if staged else inp
    outs = list(h_out.split(out_splits, dim=0))
    ins = list(h_inp.split(in_splits, dim=0))
    outs[rank].copy_(ins[rank])
    reqs = []
    for peer in range(world):
        if peer == rank:
            continue
        g_peer = dist.get_global_rank(group, peer) if group is not dist.group.WORLD else peer
        if in_splits[peer] > 0:
            reqs.append(dist.isend(ins[peer].contiguous(), g_peer, group=group))
        if out_splits[peer] > 0:
            reqs.append(dist.irecv(outs[peer], g_peer, group=group))
    for r in reqs:
        r.wait()
    if staged:
        out.copy_(h_out)


def _allgather_rows(out: Tensor, inp: Tensor, group) -> None:
    """``out[r * n : (r + 1) * n] <- inp`` of rank ``r``; host-staged when the backend cannot move device memory."""
    if _backend(group) == "sim":
        out.view(-1, *inp.shape).copy_(inp.unsqueeze(0).expand(_world(group), *inp.shape))
        return
    if _backend(group) == "nccl" or not inp.is_cuda:
        dist.all_gather_into_tensor(out, inp, group=group)
        return
    h_out = torch.empty(out.shape, dtype=out.dtype)
    dist.all_gather_into_tensor(h_out, inp.cpu(), group=group)
    out.copy_(h_out)


@dataclass
class HaloExchange:
    """Send / receive lists of one all-to-all-v of boundary rows (indices local to the owner's row range)."""

    send_idx: Tensor  # int64 [n_send]  own-row indices, grouped by destination rank (ascending)
    send_splits: List[int]
    recv_splits: List[int]
    group: object
    # send buffers by (dtype, width), allocated once: the same exchange runs in every block of every step, and a block's
    # pack is ordered behind the previous block's transfer (``finish`` lets the stream wait for it) -- one buffer is enough
    _send: dict = field(default_factory=dict, repr=False, compare=False)
    # measurement hook (bench.py ``exchanges``): a list collects, per exchange, four device events on the launch stream --
    # before / after ``start`` (pack + enqueue) and before / after ``finish`` (the stream waiting for the transfer)
    TIMING = None

    @property
    def n_recv(self) -> int:
        return sum(self.recv_splits)

    def _pack(self, rows: Tensor, n_own: int) -> Tensor:
        """The rows the other ranks need, packed in destination-rank order (one gather kernel) into the resident buffer."""
        key = (rows.dtype, rows.shape[1], rows.device)
        buf = self._send.get(key)
        if buf is None:
            buf = self._send[key] = torch.empty((self.send_idx.shape[0], rows.shape[1]), dtype=rows.dtype,
                                                device=rows.device)
        return torch.index_select(rows[:n_own], 0, self.send_idx, out=buf)

    def start(self, rows: Tensor, n_own: int):
        """Begin ``rows[n_own : n_own + n_recv] <-`` the rows this rank's halo needs (``rows[:n_own]`` are its own).

        Returns a handle for :meth:`finish`; between the two the caller may launch work that does not touch the halo
        rows (the x_r | q | u GEMM), which overlaps with the xGMI transfer."""
        marks = None
        if HaloExchange.TIMING is not None and rows.is_cuda:
            marks = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            marks[0].record()
        send = self._pack(rows, n_own) if not (torch.is_grad_enabled() and rows.requires_grad) else \
            rows[:n_own].index_select(0, self.send_idx)
        work = _alltoallv(rows[n_own:n_own + self.n_recv], send, self.recv_splits, self.send_splits, self.group,
                          async_op=True)
        if marks is not None:
            marks[1].record()
        return (work, send, marks)  # keep the send buffer alive until the transfer has been waited for

    @staticmethod
    def finish(handle) -> None:
        work, marks = handle[0], handle[2]
        if marks is not None:
            marks[2].record()
        if work is not None:
            work.wait()  # the current stream waits for RCCL's stream; the host does not block
        if marks is not None:
            marks[3].record()
            if HaloExchange.TIMING is not None:
                HaloExchange.TIMING.append(tuple(marks))

    def exchange(self, rows: Tensor, n_own: int) -> None:
        self.finish(self.start(rows, n_own))


@dataclass
class HeadExchange:
    """rows <-> heads all-to-all around global attention (the reference's Ulysses scheme, distributed/transformer.py):
    rank ``r`` owns ``rows[r]`` mesh rows everywhere else and heads ``tensor_split(H, P)[r]`` inside the attention."""

    rows: List[int]  # mesh rows owned by every rank
    rank: int
    group: object
    # the ranks own contiguous ranges of the INTERNAL (Morton) mesh order; a sliding attention window acts on the EXTERNAL
    # node order, so the full sequence is permuted around the windowed attention: internal position of every external
    # row / external row of every internal position
    to_external: Optional[Tensor] = None
    to_internal: Optional[Tensor] = None

    def _head_bounds(self, num_heads: int) -> List[int]:
        return split_bounds(num_heads, len(self.rows))

    def local_heads(self, num_heads: int) -> int:
        b = self._head_bounds(num_heads)
        return b[self.rank + 1] - b[self.rank]

    def rows_to_heads(self, qkv: Tensor, num_heads: int) -> Tensor:
        """``[n_own, 3C]`` (q|k|v, all heads) -> ``[S, 3 * C_local]`` (q|k|v of this rank's heads, all rows)."""
        n_own, c3 = qkv.shape
        d = c3 // 3 // num_heads
        hb = self._head_bounds(num_heads)
        h_loc = hb[self.rank + 1] - hb[self.rank]
        if h_loc == 0:
            raise NotImplementedError("more ranks than attention heads")
        q4 = qkv.view(n_own, 3, num_heads, d)
        send = torch.cat([q4[:, :, hb[p]:hb[p + 1], :].reshape(-1) for p in range(len(self.rows))])
        in_splits = [n_own * 3 * (hb[p + 1] - hb[p]) * d for p in range(len(self.rows))]
        out_splits = [r * 3 * h_loc * d for r in self.rows]
        recv = torch.empty(sum(out_splits), dtype=qkv.dtype, device=qkv.device)
        _alltoallv(recv, send, out_splits, in_splits, self.group)
        return recv.view(sum(self.rows), 3 * h_loc * d)  # rows of all ranks in rank order, each [3, h_loc, d]

    def heads_to_rows(self, att: Tensor, num_heads: int) -> Tensor:
        """``[S, C_local]`` -> ``[n_own, C]``."""
        s_len, c_loc = att.shape
        hb = self._head_bounds(num_heads)
        h_loc = hb[self.rank + 1] - hb[self.rank]
        d = c_loc // h_loc
        n_own = self.rows[self.rank]
        in_splits = [r * c_loc for r in self.rows]
        out_splits = [n_own * (hb[p + 1] - hb[p]) * d for p in range(len(self.rows))]
        recv = torch.empty(sum(out_splits), dtype=att.dtype, device=att.device)
        _alltoallv(recv, att.reshape(-1), out_splits, in_splits, self.group)
        parts = [blk.view(n_own, hb[p + 1] - hb[p], d) for p, blk in enumerate(recv.split(out_splits))]
        return torch.cat(parts, dim=1).reshape(n_own, num_heads * d)


@dataclass
class LocalGraph:
    plan: EdgePlan  # CSR over LOCAL indices; plan.perm holds ORIGINAL edge ids (for the attribute gather)
    n_own_src: int  # leading source rows that are this rank's own (the rest is halo)
    halo: Optional[HaloExchange]
    heads: Optional[HeadExchange] = None


def _owner(ids: Tensor, bounds: Tensor) -> Tensor:
    return torch.bucketize(ids, bounds[1:], right=True)


def _local_graph(src: Tensor, dst_local: Tensor, e_ids: Tensor, n_src: int, n_dst: int) -> EdgePlan:
    plan = runtime.build_edge_plan(torch.stack([src, dst_local]), n_src, n_dst)
    plan.perm = e_ids[plan.perm.long()].to(torch.int32)
    return plan


def _halo_lists(src_int: Tensor, dst_owner_mask_fn, bounds: List[int], rank: int, world: int, group) -> tuple:
    """For mesh-source edge sets: (halo ids of `rank`, HaloExchange).  ``dst_owner_mask_fn(p)`` -> edge mask of rank p."""
    dev = src_int.device
    b = torch.tensor(bounds, device=dev)
    need = []  # need[p] = sorted unique mesh ids (internal) that rank p reads but does not own
    for p in range(world):
        s = src_int[dst_owner_mask_fn(p)]
        s = torch.unique(s[(s < bounds[p]) | (s >= bounds[p + 1])])
        need.append(s)
    mine = need[rank]
    recv_splits = [int(((mine >= bounds[p]) & (mine < bounds[p + 1])).sum()) for p in range(world)]
    send_parts, send_splits = [], []
    for p in range(world):
        s = need[p]
        s = s[(s >= bounds[rank]) & (s < bounds[rank + 1])] - bounds[rank]
        send_parts.append(s)
        send_splits.append(int(s.numel()))
    _ = b
    return mine, HaloExchange(torch.cat(send_parts), send_splits, recv_splits, group)


@dataclass
class ShardPlan:
    rank: int
    world: int
    lo: int
    hi: int
    enc_src_ids: Tensor  # grid ids (ascending) whose embeddings this rank needs
    enc: LocalGraph
    proc: LocalGraph
    dec_dst_ids: Tensor  # grid ids (ascending) this rank produces
    dec: LocalGraph
    dec_counts: List[int]  # rows produced by every rank
    dec_all_ids: Tensor  # concatenation of every rank's dec_dst_ids (rank order)
    gather_pos: Optional[Tensor] = None  # per grid row: its position in the padded all-gather buffer (built on first use)
    io_rows: Optional[Tensor] = None  # cat[enc_src_ids, dec_dst_ids]: the grid rows whose input features this rank assembles
    # autoregressive rollout with the state kept sharded: the grid rows this rank's ENCODER reads but another rank
    # decodes (grid ids, grouped by owner) and the exchange that fetches their predictions (send_idx = positions in
    # dec_dst_ids)
    grid_halo_ids: Optional[Tensor] = None
    grid_halo: Optional[HaloExchange] = None


def build_shard_plan(model, group, device) -> ShardPlan:
    rank, world = _rank(group), _world(group)
    order, inv = model._mesh_order(device)
    n_mesh = order.shape[0]
    bounds = split_bounds(n_mesh, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    n_own = hi - lo
    bt = torch.tensor(bounds, device=device)

    # ---- encoder: grid -> mesh, destinations = own mesh rows, sources gathered locally
    ei = model.encoder.edge_index_base
    dst_int = inv[ei[1]]
    e_ids = torch.nonzero((dst_int >= lo) & (dst_int < hi)).flatten()
    src = ei[0][e_ids]
    enc_src_ids = torch.unique(src)
    enc = LocalGraph(_local_graph(torch.searchsorted(enc_src_ids, src), dst_int[e_ids] - lo, e_ids,
                                  int(enc_src_ids.numel()), n_own), int(enc_src_ids.numel()), None)

    # ---- processor: mesh -> mesh (the Transformer processor has no edges: rows <-> heads exchange instead)
    heads = HeadExchange([bounds[p + 1] - bounds[p] for p in range(world)], rank, group, inv, order)
    if not hasattr(model.processor, "edge_index_base"):
        proc = LocalGraph(None, n_own, None, heads)
    ei = getattr(model.processor, "edge_index_base", None)
    if ei is None:
        ei = torch.zeros((2, 0), dtype=torch.int64, device=device)
    src_int, dst_int = inv[ei[0]], inv[ei[1]]
    # rank r's edges = chunk r of sort_edges_1hop_chunks over the Morton-relabelled destinations (original edge order)
    dst_owner = edge_chunk_ids(dst_int, n_mesh, world)
    halo_ids, halo = _halo_lists(src_int, lambda p: dst_owner == p, bounds, rank, world, group)
    e_order, e_counts = partition_edges_by_dst(dst_int, n_mesh, world)
    e_ids = e_order[sum(e_counts[:rank]):sum(e_counts[:rank + 1])]
    s = src_int[e_ids]
    own = (s >= lo) & (s < hi)
    s_local = torch.where(own, s - lo, n_own + torch.searchsorted(halo_ids, s))
    if hasattr(model.processor, "edge_index_base"):
        proc = LocalGraph(_local_graph(s_local, dst_int[e_ids] - lo, e_ids, n_own + int(halo_ids.numel()), n_own),
                          n_own, halo, heads)

    # ---- decoder: mesh -> grid; a grid node goes to the owner of its first mesh source
    ei = model.decoder.edge_index_base
    n_grid = int(model.node_attributes.num_nodes[model._graph_name_data])
    src_int, dst = inv[ei[0]], ei[1]
    full = runtime.build_edge_plan(torch.stack([src_int, dst]), n_mesh, n_grid)  # stable: first CSR slot = first edge
    has_edge = full.rowptr[1:] > full.rowptr[:-1]
    first_slot = full.rowptr[:-1].long().clamp_max(max(ei.shape[1] - 1, 0))
    first_src = torch.where(has_edge, full.col.long()[first_slot] if ei.shape[1] > 0 else first_slot,
                            torch.full_like(first_slot, -1))
    g_owner = torch.where(first_src >= 0, _owner(first_src.clamp_min(0), bt),
                          torch.arange(n_grid, device=device) % world)
    e_owner = g_owner[dst]
    halo_ids, halo = _halo_lists(src_int, lambda p: e_owner == p, bounds, rank, world, group)
    dec_dst_ids = torch.nonzero(g_owner == rank).flatten()
    e_ids = torch.nonzero(e_owner == rank).flatten()
    s = src_int[e_ids]
    own = (s >= lo) & (s < hi)
    s_local = torch.where(own, s - lo, n_own + torch.searchsorted(halo_ids, s))
    dec = LocalGraph(_local_graph(s_local, torch.searchsorted(dec_dst_ids, dst[e_ids]), e_ids,
                                  n_own + int(halo_ids.numel()), int(dec_dst_ids.numel())), n_own, halo)
    counts = torch.bincount(g_owner, minlength=world).tolist()
    all_ids = torch.argsort(g_owner, stable=True)  # rank-major, ascending grid id inside a rank
    # ---- grid halo of the rollout: encoder sources of rank p that rank p does not decode itself
    ei = model.encoder.edge_index_base
    e_rank = _owner(inv[ei[1]], bt)  # rank whose mesh row an encoder edge feeds
    need = []
    for p in range(world):
        s = torch.unique(ei[0][e_rank == p])
        need.append(s[g_owner[s] != p])
    mine = need[rank]
    mine = mine[torch.argsort(g_owner[mine], stable=True)]  # grouped by the rank that decodes them
    recv_splits = torch.bincount(g_owner[mine], minlength=world).tolist()
    send_parts = [torch.searchsorted(dec_dst_ids, need[p][g_owner[need[p]] == rank]) for p in range(world)]
    grid_halo = HaloExchange(torch.cat(send_parts), [int(t.numel()) for t in send_parts], recv_splits, group)
    return ShardPlan(rank, world, lo, hi, enc_src_ids, enc, proc, dec_dst_ids, dec, counts, all_ids,
                     grid_halo_ids=mine, grid_halo=grid_halo)


def gather_output_rows(sp: "ShardPlan", y_local: Tensor, group, grid: int) -> Tensor:
    """All-gather (padded to the largest shard) of the per-rank output rows, put back into grid order: ``[grid, V_out]``."""
    v_out = y_local.shape[1]
    max_rows = max(sp.dec_counts)
    send = torch.zeros((max_rows, v_out), dtype=torch.float32, device=y_local.device)
    send[: y_local.shape[0]] = y_local
    gathered = torch.empty((sp.world * max_rows, v_out), dtype=torch.float32, device=y_local.device)
    _allgather_rows(gathered, send, group)
    return gathered.index_select(0, _gather_positions(sp, grid, y_local.device))


def sharded_forward(model, x: Tensor, group, input_affine=None, output_affine=None, local_output: bool = False):
    """Full-input / full-output forward with the mesh partitioned over ``group`` (batch size 1, as in the reference).

    ``local_output=True`` (rollout with the state kept sharded, :func:`advance_sharded_state`): the final all-gather is
    skipped and ``(y_local [rows, V_out] f32 WITHOUT the prognostic residual, shard plan)`` is returned; of ``x`` only
    the grid rows this rank's encoder and decoder read (``enc_src_ids``, ``dec_dst_ids``) need to be valid."""
    if torch.is_grad_enabled() and any(p.requires_grad for p in model.parameters()):
        if input_affine is not None or output_affine is not None:
            raise NotImplementedError("input_affine / output_affine belong to the inference interface (predict_step)")
        return sharded_training_forward(model, x, group)
    batch_size, _, ensemble_size, grid, _ = x.shape
    assert batch_size == 1, "Only batch size of 1 is supported when model is sharded across GPUs"
    if ensemble_size != 1:
        # (the reference shards the '(batch ensemble grid)' rows, models/encoder_processor_decoder.py:173-186, so ensembles
        # run there; here the plan's grid-row ids address ONE member -- tracked as a gap, refused before any plan is built)
        raise NotImplementedError("the node-partitioned forward runs ensemble size 1 (batch 1 per model group, as the reference)")
    dtype = runtime.compute_dtype(x)
    from ..layers.mapper import GNNBaseMapper

    if not hasattr(model.encoder, "native_local") or not hasattr(model.decoder, "native_local"):
        raise NotImplementedError("the node-partitioned forward supports the GraphTransformer and GNN mappers")
    gnn_maps = isinstance(model.encoder, GNNBaseMapper)
    if gnn_maps != isinstance(model.decoder, GNNBaseMapper):
        raise NotImplementedError("the node-partitioned forward needs encoder and decoder of one mapper family")
    if not gnn_maps and model.encoder.proc.fold_width(dtype) is None:
        raise NotImplementedError("the node-partitioned forward needs the folded edge kernel for this shape / dtype")
    kmult = ops.k_multiple(dtype)
    key = ("shard_plan", str(x.device), _world(group), _rank(group))
    if key not in model._idx_cache:
        model._idx_cache[key] = build_shard_plan(model, group, x.device)
    sp: ShardPlan = model._idx_cache[key]
    data, hidden = model._graph_name_data, model._graph_name_hidden
    na = model.node_attributes
    order, _ = model._mesh_order(x.device)
    own_ids = order[sp.lo:sp.hi]

    # [x | coordinates | trainable | 1 | 0-pad] as in the single-device forward (the constant 1 carries the embedding bias
    # of the GraphTransformer mappers' embedding fold; every rank assembles the full grid and selects its rows)
    fold = model._embed_fold(dtype)
    width = model.multi_step * model.num_input_channels + na.attr_ndims[data]
    # ... of the grid rows this rank's encoder reads and its decoder writes only (about 1 / world of the grid each), not of
    # the whole grid: the full assembly is 0.3 ms at N320 whatever the group size
    if sp.io_rows is None:
        sp.io_rows = torch.cat([sp.enc_src_ids, sp.dec_dst_ids])
    n_enc = sp.enc_src_ids.shape[0]
    x_rows = ops.assemble_nodes(x, na.latlons(data), model._with_ones(na.trainable_tensors[data].trainable, grid, fold),
                                1, dtype, ld_out=model._feature_ld(width + int(fold), dtype, fold), in_affine=input_affine,
                                rows=sp.io_rows)
    x_enc_src, x_dec_rows = x_rows[:n_enc], x_rows[n_enc:]
    tr_hidden = na.trainable_tensors[hidden].trainable
    w_hidden = na.attr_ndims[hidden]
    x_hidden = ops.assemble_nodes(None, na.latlons(hidden)[own_ids],
                                  model._with_ones(None if tr_hidden is None else tr_hidden[own_ids], own_ids.numel(), fold,
                                                   device=x.device),
                                  1, dtype, ld_out=model._feature_ld(w_hidden + int(fold), dtype, fold))
    one_data, one_hidden = (width, w_hidden) if fold else (None, None)

    if gnn_maps:
        # GNN mappers hand the UPDATED grid embedding on to the decoder (reference layers/mapper.py:522): the rows this
        # rank decodes are embedded and updated here, next to the rows that feed its mesh nodes (both row-local)
        x_dec_dst, x_latent = model.encoder.native_local(x_enc_src, x_hidden, sp.enc, x_src_extra=x_dec_rows)
    else:
        x_latent = model.encoder.native_local(x_enc_src, x_hidden, sp.enc, one_cols=(one_data, one_hidden))
        x_dec_dst = x_dec_rows
    x_proc = model.processor.native_local(x_latent, sp.proc)
    x_latent_proc = ops.add(x_proc, x_latent)
    if gnn_maps:
        y_local = model.decoder.native_local(x_latent_proc, x_dec_dst, sp.dec, out_dtype=torch.float32)
    else:
        y_local = model.decoder.native_local(x_latent_proc, x_dec_dst, sp.dec, out_dtype=torch.float32,
                                             one_cols=(None, one_data))
    if isinstance(y_local, tuple):
        y_local = y_local[1]

    if local_output:
        return y_local, sp
    # residual, boundings and de-normalisation are row-local: every rank finishes the rows it decoded, then they are gathered
    y_local = model._finish(y_local.float().view(1, 1, -1, model.num_output_channels), x, input_affine, output_affine,
                            rows=sp.dec_dst_ids)
    y = gather_output_rows(sp, y_local.view(-1, model.num_output_channels), group, grid)
    return y.view(1, ensemble_size, grid, model.num_output_channels).to(y_local.dtype)


# ------------------------------------------------------------------------------------------ training (backward collectives)
class _HaloRows(torch.autograd.Function):
    """``[own rows | halo rows]`` of a row matrix: forward = the halo all-to-all-v of :class:`HaloExchange`; backward = the
    REVERSE all-to-all-v (the gradients of the halo copies travel back to the ranks that own the rows, counts swapped)
    followed by an index-add into the owners' rows.  The counterpart of the reference's autograd-wrapped collectives
    (reference distributed/graph.py:152-162, distributed/transformer.py:144-152), moving O(boundary) rows instead of whole
    tensors."""

    @staticmethod
    def forward(ctx, rows_own: Tensor, halo: HaloExchange) -> Tensor:
        n_own = rows_own.shape[0]
        full = torch.empty((n_own + halo.n_recv, rows_own.shape[1]), dtype=rows_own.dtype, device=rows_own.device)
        full[:n_own].copy_(rows_own)
        halo.exchange(full, n_own)
        ctx.halo, ctx.n_own = halo, n_own
        return full

    @staticmethod
    def backward(ctx, dfull: Tensor):
        halo, n_own = ctx.halo, ctx.n_own
        d_own = dfull[:n_own].clone()
        n_send = sum(halo.send_splits)
        # the reverse all-to-all-v is a GROUP-WIDE collective (RCCL all_to_all_single): a rank with an empty halo enters it
        # with zero-length splits exactly as HaloExchange.start does in the forward -- skipping it would leave its peers
        # waiting in (or mis-pair) the collective.  Only the accumulation is conditional.
        back = torch.empty((n_send, dfull.shape[1]), dtype=dfull.dtype, device=dfull.device)
        _alltoallv(back, dfull[n_own:].contiguous(), halo.send_splits, halo.recv_splits, halo.group)
        if n_send > 0:
            d_own.index_add_(0, halo.send_idx, back)
        return d_own, None


class _GatherOutput(torch.autograd.Function):
    """Padded all-gather of the per-rank output rows and their placement in grid order; backward = this rank's rows of the
    gradient (every rank of the model group evaluates the same loss on the same full output, so the slice -- not a
    reduction -- is the gradient: the reference's ``gather_tensor``, distributed/graph.py:60-91)."""

    @staticmethod
    def forward(ctx, y_local: Tensor, sp: "ShardPlan", group, grid: int) -> Tensor:
        v_out = y_local.shape[1]
        max_rows = max(sp.dec_counts)
        send = torch.zeros((max_rows, v_out), dtype=y_local.dtype, device=y_local.device)
        send[: y_local.shape[0]] = y_local
        gathered = torch.empty((sp.world * max_rows, v_out), dtype=y_local.dtype, device=y_local.device)
        _allgather_rows(gathered, send, group)
        ctx.ids = sp.dec_dst_ids
        return gathered.index_select(0, _gather_positions(sp, grid, y_local.device))

    @staticmethod
    def backward(ctx, dy: Tensor):
        return dy.index_select(0, ctx.ids), None, None, None


def _gather_positions(sp: "ShardPlan", grid: int, device) -> Tensor:
    if sp.gather_pos is None:  # grid order <- (rank, row) order of the padded buffer
        max_rows = max(sp.dec_counts)
        starts = torch.tensor([r * max_rows for r in range(sp.world)], device=device)
        counts = torch.tensor(sp.dec_counts, device=device)
        offs = torch.cumsum(counts, 0) - counts
        rank_of = torch.repeat_interleave(torch.arange(sp.world, device=device), counts)
        pos = starts[rank_of] + torch.arange(int(counts.sum()), device=device) - offs[rank_of]
        gp = torch.empty(grid, dtype=torch.long, device=device)
        gp[sp.dec_all_ids] = pos
        sp.gather_pos = gp
    return sp.gather_pos


def sharded_training_forward(model, x: Tensor, group) -> Tensor:
    """The node-partitioned forward WITH an autograd graph (flat model, GraphTransformer or GNN mappers around a
    GraphTransformer or GNN processor, batch size 1): the same partition
    as :func:`sharded_forward`, every block on the differentiable kernels of ``autograd.py``, the halo exchanges and the
    output gather as autograd functions with their backward collectives.  Each rank ends up with the gradient
    contributions of ITS rows for every parameter: the caller sums them over the model group (what anemoi-training's DDP
    strategy does across all ranks of a model instance)."""
    from .. import autograd
    from .. import training
    from ..layers.mapper import GNNBaseMapper
    from ..layers.mapper import GraphTransformerBaseMapper
    from ..layers.processor import GNNProcessor
    from ..layers.processor import GraphTransformerProcessor

    gt_maps = isinstance(model.encoder, GraphTransformerBaseMapper) and isinstance(model.decoder, GraphTransformerBaseMapper)
    gnn_maps = isinstance(model.encoder, GNNBaseMapper) and isinstance(model.decoder, GNNBaseMapper)
    from ..layers.processor import TransformerProcessor

    gt_proc, gnn_proc = isinstance(model.processor, GraphTransformerProcessor), isinstance(model.processor, GNNProcessor)
    tfm_proc = isinstance(model.processor, TransformerProcessor)
    if not ((gt_maps or gnn_maps) and (gt_proc or gnn_proc or tfm_proc)):
        raise NotImplementedError("node-partitioned training: GraphTransformer or GNN mappers (one family for both) around "
                                  "a GraphTransformer, GNN or Transformer processor")
    b, _, ens, grid, _ = x.shape
    assert b == 1, "Only batch size of 1 is supported when model is sharded across GPUs"
    if ens != 1:
        raise NotImplementedError("node-partitioned training: ensemble size 1")
    dtype = runtime.compute_dtype(x)
    key = ("shard_plan", str(x.device), _world(group), _rank(group))
    if key not in model._idx_cache:
        model._idx_cache[key] = build_shard_plan(model, group, x.device)
    sp: ShardPlan = model._idx_cache[key]
    data, hidden = model._graph_name_data, model._graph_name_hidden
    order, _ = model._mesh_order(x.device)
    own_ids = order[sp.lo:sp.hi]
    enc, proc, dec = model.encoder, model.processor, model.decoder
    heads = proc.proc[0].blocks[0].num_heads if gt_proc else (enc.proc.num_heads if gt_maps else 1)
    if tfm_proc and proc.proc[0].blocks[0].attention.num_heads % sp.world != 0:
        # (as the reference: its _headsalltoall / _seqalltoall size every receive buffer like the rank's own piece,
        #  distributed/transformer.py:41-52,77 -- equal head shares; the inference route's HeadExchange takes uneven ones)
        raise NotImplementedError("node-partitioned training of the Transformer processor needs heads divisible by the group size")
    if gt_proc or gt_maps:
        training._check_heads(model.num_channels, heads, dtype)

    def gnn_attrs(mod, plan):  # the plain attribute matrix [edge_attr | trainable] in the local plan's CSR order
        parts = [mod.edge_attr.float()] + ([] if mod.trainable.trainable is None else [mod.trainable.trainable.float()])
        return training._cast(torch.cat(parts, dim=1).index_select(0, plan.perm.long()), dtype)  # (this rank's edges only)

    def gnn_encoder(src_rows, extra_rows, x_hid):
        """GNNForwardMapper on the local graph (all sources local): (updated embedding of ``extra_rows`` -- the grid rows
        this rank decodes, reference layers/mapper.py:522 --, new mesh rows)."""
        e = training.mlp(enc.emb_edges, gnn_attrs(enc, sp.enc.plan))
        hs, hd = training.mlp(enc.emb_nodes_src, src_rows), training.mlp(enc.emb_nodes_dst, x_hid)
        (_, hd), _ = training.gnn_mapper_block_csr(enc.proc, hs, hd, e, sp.enc.plan)
        hx = training.mlp(enc.emb_nodes_src, extra_rows)
        if enc.proc.update_src_nodes:  # row-local update of the source embedding (reference layers/block.py:282)
            hx = training.mlp(enc.proc.node_mlp, torch.cat([hx, hx], dim=1), residual=hx)
        return hx, hd

    def gnn_decoder(h_mesh_own, h_grid_own):
        e = training.mlp(dec.emb_edges, gnn_attrs(dec, sp.dec.plan))
        src = _HaloRows.apply(h_mesh_own, sp.dec.halo) if sp.dec.halo is not None else h_mesh_own
        (_, hd), _ = training.gnn_mapper_block_csr(dec.proc, src, h_grid_own, e, sp.dec.plan)
        return training.mlp(dec.node_data_extractor, hd)

    def gnn_processor_chunk(chunk, h, e):
        if chunk.emb_edges is not None:
            e = training.mlp(chunk.emb_edges, e)
        for blk in chunk.blocks:  # GraphConvProcessorBlock: the sources are the own rows + the halo rows of their owners
            src = _HaloRows.apply(h, sp.proc.halo) if sp.proc.halo is not None else h
            e = training._gnn_edge_update(blk.conv.edge_mlp, h, src, e, sp.proc.plan)
            agg = autograd.segment_sum(e, sp.proc.plan)
            h = training.mlp(blk.node_mlp, torch.cat([h, agg], dim=1), residual=h)
        return h, e

    def attrs(mod, plan):
        return autograd._edge_attr_csr(mod.edge_attr, mod.trainable.trainable, plan, ops.round_up(mod.edge_dim + 1, 4))

    def mapper_block(mod, h_src_own, h_dst, lg: LocalGraph):
        """GraphTransformerMapperBlock on a local graph: keys / values of the own source rows, halo rows appended."""
        blk, sd = mod.proc, training._block_sd(mod.proc)
        g = lambda name: sd["b." + name]  # noqa: E731
        c = h_dst.shape[1]
        up = ops.round_up(mod.edge_dim + 1, 4)
        w_u, b_u, w_t = autograd._lin_edge_fold(sd, "b", c, heads, up, h_dst.device)
        xs = autograd.layer_norm(h_src_own, g("layer_norm1.weight"), g("layer_norm1.bias"), blk.layer_norm1.eps)
        xd = autograd.layer_norm(h_dst, g("layer_norm2.weight"), g("layer_norm2.bias"), blk.layer_norm2.eps)
        kv = autograd.linear(xs, torch.cat([g("lin_key.weight"), g("lin_value.weight")], 0),
                             torch.cat([g("lin_key.bias"), g("lin_value.bias")], 0))
        if lg.halo is not None:
            kv = _HaloRows.apply(kv, lg.halo)
        sq = autograd.linear(xd, torch.cat([g("lin_self.weight"), g("lin_query.weight"), w_u], 0),
                             torch.cat([g("lin_self.bias"), g("lin_query.bias"), b_u], 0))
        att = autograd.gt_edge_attention_packed(sq, kv, attrs(mod, lg.plan), lg.plan, heads, up)
        return autograd._gt_tail(att, h_dst, sd, "b", w_t, blk.activation, blk.layer_norm1.eps)

    def processor_block(blk, h, ea, lg: LocalGraph):
        sd = training._block_sd(blk)
        g = lambda name: sd["b." + name]  # noqa: E731
        c = h.shape[1]
        up = ea.shape[1]
        w_u, b_u, w_t = autograd._lin_edge_fold(sd, "b", c, heads, up, h.device)
        xh = autograd.layer_norm(h, g("layer_norm1.weight"), g("layer_norm1.bias"), blk.layer_norm1.eps)
        kv = _HaloRows.apply(autograd.linear(xh, torch.cat([g("lin_key.weight"), g("lin_value.weight")], 0),
                                             torch.cat([g("lin_key.bias"), g("lin_value.bias")], 0)), lg.halo)
        sq = autograd.linear(xh, torch.cat([g("lin_self.weight"), g("lin_query.weight"), w_u], 0),
                             torch.cat([g("lin_self.bias"), g("lin_query.bias"), b_u], 0))
        att = autograd.gt_edge_attention_packed(sq, kv, ea, lg.plan, heads, up)
        return autograd._gt_tail(att, h, sd, "b", w_t, blk.activation, blk.layer_norm1.eps)

    with torch.autocast(device_type=x.device.type, enabled=False):
        x_data = torch.cat([x.permute(0, 2, 3, 1, 4).reshape(grid, -1), training._node_rows(model, data, 1)], dim=1).to(dtype)
        x_hidden = training._node_rows(model, hidden, 1).index_select(0, own_ids).to(dtype)
        # encoder: the grid rows that feed this rank's mesh rows -- no communication (every rank holds the full input)
        if gt_maps:
            hs = autograd.linear(x_data.index_select(0, sp.enc_src_ids), enc.emb_nodes_src.weight, enc.emb_nodes_src.bias)
            hd = autograd.linear(x_hidden, enc.emb_nodes_dst.weight, enc.emb_nodes_dst.bias)
            x_latent = training._checkpoint(lambda a, c_: mapper_block(enc, a, c_, sp.enc), hs, hd)
            x_dec_dst = None
        else:
            x_dec_dst, x_latent = training._checkpoint(gnn_encoder, x_data.index_select(0, sp.enc_src_ids),
                                                       x_data.index_select(0, sp.dec_dst_ids), x_hidden)
        # processor: one halo exchange per block (k|v rows / node rows), forward and (reversed) backward
        if gt_proc:
            ea = attrs(proc, sp.proc.plan)

            def run_chunk(chunk, h, ea_):
                for blk in chunk.blocks:
                    h = processor_block(blk, h, ea_, sp.proc)
                return h

            h = x_latent
            for chunk in proc.proc:
                h = training._checkpoint(run_chunk, chunk, h, ea)
        elif gnn_proc:
            h, e = x_latent, gnn_attrs(proc, sp.proc.plan)
            for chunk in proc.proc:
                h, e = training._checkpoint(gnn_processor_chunk, chunk, h, e)
        else:
            # Transformer processor: the modules' own sequence-sharded route (rows <-> heads exchanges as autograd nodes,
            # layers/attention.py::_sharded).  Global attention does not care which rows a rank holds; a sliding window is
            # defined on the external node order, which the Morton-ordered partition does not keep.
            if proc.proc[0].blocks[0].attention.attention_window() >= 0:
                raise NotImplementedError("node-partitioned training with a sliding attention window")
            rows_of = sp.proc.heads.rows
            h = proc(x_latent, 1, [[r_, x_latent.shape[1]] for r_ in rows_of], group)
        x_latent_proc = h + x_latent
        # decoder: own grid rows as destinations, own + halo mesh rows as sources
        if gt_maps:
            hd = autograd.linear(x_data.index_select(0, sp.dec_dst_ids), dec.emb_nodes_dst.weight, dec.emb_nodes_dst.bias)
            y_local = training._checkpoint(lambda a, c_: mapper_block(dec, a, c_, sp.dec), x_latent_proc, hd, last=True)
            y_local = training.sequential(dec.node_data_extractor, y_local).float()
        else:
            y_local = training._checkpoint(gnn_decoder, x_latent_proc, x_dec_dst, last=True).float()
        y = _GatherOutput.apply(y_local, sp, group, grid)
        return training._finish(model, y, x, 1, 1, grid)


def finish_local_rows(model, x_state: Tensor, y_local: Tensor, sp: ShardPlan) -> Tensor:
    """Prognostic residual (from the last time slice of this rank's rows of the state) and boundings on the rows this
    rank decodes: the row-local part of ``AnemoiModelEncProcDec._finish`` (reference :227-231)."""
    assert x_state.shape[0] == 1 and x_state.shape[2] == 1, "a sharded state is batch 1, ensemble 1 (reference layers/block.py:501-504)"
    out_idx, in_idx = model._prognostic_indices(x_state.device)
    own = y_local.float().clone()
    own[:, out_idx.long()] += x_state[0, -1, 0].index_select(0, sp.dec_dst_ids)[:, in_idx.long()]
    if len(model.boundings) > 0:
        plan = model._bounding_plan(x_state.device, None)
        if plan is None:
            for bounding in model.boundings:
                own = bounding(own)
        else:
            ops.bound_output(own, *plan[0])
    return own


def sharded_state_output(model, x_state: Tensor, y_local: Tensor, sp: ShardPlan, group) -> Tensor:
    """The full prediction ``[1, Ens, grid, V_out]`` of a step computed on a sharded state: rows finished locally, then
    all-gathered (the residual must come from the rank that holds the row's state)."""
    _, _, ens, grid, _ = x_state.shape
    return gather_output_rows(sp, finish_local_rows(model, x_state, y_local, sp), group, grid).view(1, ens, grid, -1)


def advance_sharded_state(model, x_state: Tensor, y_local: Tensor, sp: ShardPlan, colmap: Tensor,
                          forcing: Optional[Tensor] = None) -> Tensor:
    """One autoregressive step on a SHARDED state (no per-step all-gather): ``y_local`` are this rank's
    decoded grid rows of the NORMALISED prediction without the prognostic residual (``sharded_forward(local_output=True)``).
    The rows are finished locally (residual, boundings), the rows of other ranks that this rank's encoder reads arrive by
    ONE all-to-all-v of ``[rows, V_out]`` f32 (the grid halo: a few per cent of the grid instead of all of it), and
    ``anemoi_advance_input`` shifts the time axis in place.  Afterwards exactly the rows ``enc_src_ids`` /
    ``dec_dst_ids`` of ``x_state`` are valid on this rank -- all the next ``sharded_forward`` reads."""
    b, _, ens, grid, _ = x_state.shape
    assert b == 1 and ens == 1, "a sharded state is batch 1, ensemble 1 (reference layers/block.py:501-504)"
    v_out = y_local.shape[1]
    rows = torch.cat([sp.dec_dst_ids, sp.grid_halo_ids])
    y_rows = torch.empty((rows.shape[0], v_out), dtype=torch.float32, device=x_state.device)
    n_own = sp.dec_dst_ids.shape[0]
    y_rows[:n_own].copy_(finish_local_rows(model, x_state, y_local, sp))
    sp.grid_halo.exchange(y_rows, n_own)  # predictions of the encoder-halo grid rows, from the ranks that decode them
    # rows this rank neither decodes nor reads stay ZERO (never uninitialised memory: advance_input copies every row
    # into the state, and NaN / Inf there would reach assemble_nodes and the forcing pre-processors of rollout())
    y_full = torch.zeros((b, ens, grid, v_out), dtype=torch.float32, device=x_state.device)
    y_full[0, 0].index_copy_(0, rows, y_rows)
    return ops.advance_input(x_state, y_full, colmap, forcing)
