"""Differentiable collectives over a model communication group: the operator set of the reference's
``distributed/graph.py`` (:19-137) and ``distributed/transformer.py`` (:85-130) -- ``shard_tensor``, ``gather_tensor``,
``reduce_tensor``, ``sync_tensor``, ``reduce_shard_tensor``, ``shard_heads``, ``shard_sequence`` with the same arguments,
the same forward / backward pairing and the same identity behaviour without a group.

The model forward of this package does NOT run on them: it partitions the mesh once and exchanges halo rows
(``partition.py``).  They exist for code written against the reference's operators (a loss that gathers its shards, a
custom block that reshards between layers): same names, same results, on ``torch.distributed`` -- RCCL on MI355X
(``backend="nccl"``), gloo in the CPU tests.

Every operator is one autograd node built from three primitives, each a single flat collective so that RCCL sees one
large transfer instead of a list of small ones:

* ``take``    -- this rank's slice of a dimension (no communication),
* ``collect`` -- all ranks' slices concatenated along a dimension: shards of unequal length travel zero-padded to the
  longest through ONE ``all_gather_into_tensor``,
* ``total``   -- the sum over the ranks, accumulated in f32 whatever the dtype (as the reference reduces).

=====================  ==========================  ===============================
operator               forward                     backward
=====================  ==========================  ===============================
``shard_tensor``       take                        collect (or own slot only)
``gather_tensor``      collect                     take
``reduce_tensor``      total                       identity
``sync_tensor``        collect                     total, then take
``reduce_shard_tensor``  total, then take          collect
``shard_heads``        heads -> sequence exchange  sequence -> heads exchange
``shard_sequence``     sequence -> heads exchange  heads -> sequence exchange
=====================  ==========================  ===============================
"""

from __future__ import annotations

from typing import Callable, List, Sequence

import torch
import torch.distributed as dist
from torch import Tensor


def get_memory_format(tensor: Tensor):
    """``channels_last`` if the tensor is laid out that way, else ``contiguous_format`` (reference distributed/utils.py)."""
    if tensor.dim() == 4 and tensor.is_contiguous(memory_format=torch.channels_last) and not tensor.is_contiguous():
        return torch.channels_last
    return torch.contiguous_format


def _size(group) -> int:
    return 1 if group is None else dist.get_world_size(group=group)


def _check_dim(t: Tensor, dim: int, what: str) -> None:
    assert dim < t.dim(), f"Error, cannot {what} along {dim} for tensor with {t.dim()} dimensions."


# ------------------------------------------------------------------------------------------------------- primitives
def _take(t: Tensor, dim: int, shapes: Sequence, group) -> Tensor:
    """Rank r's slice ``shapes[r][dim]`` of dimension ``dim``."""
    if _size(group) == 1:
        return t
    _check_dim(t, dim, "split")
    fmt = get_memory_format(t)
    lengths = [int(s[dim]) for s in shapes]
    rank = dist.get_rank(group=group)
    start = sum(lengths[:rank])
    return t.narrow(dim, start, lengths[rank]).contiguous(memory_format=fmt)


def _collect(t: Tensor, dim: int, shapes: Sequence, group, communicate: bool = True) -> Tensor:
    """All ranks' shards (``shapes[r]`` each) concatenated along ``dim``.  ``communicate=False``: only this rank's slot is
    filled, the others are zeros (the reference leaves them uninitialised: a gradient nobody reads)."""
    world = _size(group)
    if world == 1:
        return t
    _check_dim(t, dim, "gather")
    fmt = get_memory_format(t)
    rank = dist.get_rank(group=group)
    lengths = [int(s[dim]) for s in shapes]
    if t.shape[dim] != lengths[rank]:
        raise ValueError(f"gather: this rank holds {t.shape[dim]} entries along {dim}, shapes say {lengths[rank]}")
    longest = max(lengths)
    front = t.movedim(dim, 0).contiguous()  # [len_r, ...]
    rest = front.shape[1:]
    if not communicate:
        out = front.new_zeros((sum(lengths), *rest))
        out.narrow(0, sum(lengths[:rank]), lengths[rank]).copy_(front)
        return out.movedim(0, dim).contiguous(memory_format=fmt)
    send = front
    if lengths[rank] != longest:
        send = front.new_zeros((longest, *rest))
        send[: lengths[rank]].copy_(front)
    recv = front.new_empty((world * longest, *rest))
    if send.is_cuda and dist.get_backend(group) != "nccl":  # (debugging ranks sharing one GPU over gloo: through the host)
        h_recv = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_gather_into_tensor(h_recv, send.cpu(), group=group)
        recv.copy_(h_recv)
    else:
        dist.all_gather_into_tensor(recv, send, group=group)
    if all(n == longest for n in lengths):
        out = recv
    else:
        out = torch.cat([recv[r * longest: r * longest + lengths[r]] for r in range(world)], dim=0)
    return out.movedim(0, dim).contiguous(memory_format=fmt)


def _total(t: Tensor, group, use_fp32: bool = True) -> Tensor:
    """Sum over the ranks (f32 accumulation by default)."""
    if _size(group) == 1:
        return t
    staged = t.is_cuda and dist.get_backend(group) != "nccl"  # (debugging ranks sharing one GPU over gloo)
    if use_fp32 and t.dtype != torch.float32:
        acc = t.float().cpu() if staged else t.float()
        dist.all_reduce(acc, group=group)
        return acc.to(device=t.device, dtype=t.dtype)
    if staged:
        acc = t.cpu().contiguous()
        dist.all_reduce(acc, group=group)
        return acc.to(t.device)
    out = t.contiguous().clone() if use_fp32 else t
    dist.all_reduce(out, group=group)
    return out


def _exchange(pieces: List[Tensor], recv_shapes: List[Sequence[int]], group) -> List[Tensor]:
    """Piece r goes to rank r, ``recv_shapes[r]`` arrives from rank r: one flat all-to-all-v."""
    from .partition import _alltoallv  # RCCL: all_to_all_single; gloo (CPU tests): point-to-point

    send = torch.cat([p.reshape(-1) for p in pieces])
    in_splits = [p.numel() for p in pieces]
    out_splits = [int(torch.Size(s).numel()) for s in recv_shapes]
    recv = send.new_empty(sum(out_splits))
    _alltoallv(recv, send, out_splits, in_splits, group)
    return [chunk.view(*shape) for chunk, shape in zip(recv.split(out_splits), recv_shapes)]


def _heads_to_sequence(t: Tensor, shapes: Sequence, group) -> Tensor:
    """``(..., H, n_local, c)`` on every rank -> ``(..., H_local, N, c)``: this rank keeps its ``tensor_split`` share of the
    heads and receives the sequence shards of all ranks (``shapes[r][0]`` rows each, rank order)."""
    world = _size(group)
    if world == 1:
        return t
    fmt = get_memory_format(t)
    rank = dist.get_rank(group=group)
    pieces = [p.contiguous() for p in torch.tensor_split(t, world, dim=-3)]
    lead, c = tuple(t.shape[:-3]), t.shape[-1]
    mine = pieces[rank].shape[-3]
    recv_shapes = [(*lead, mine, int(shapes[r][0]), c) for r in range(world)]
    return torch.cat(_exchange(pieces, recv_shapes, group), dim=-2).contiguous(memory_format=fmt)


def _sequence_to_heads(t: Tensor, shapes: Sequence, group) -> Tensor:
    """``(..., H_local, N, c)`` -> ``(..., H, n_local, c)``: the inverse exchange (the sequence is cut by ``tensor_split``, as
    the reference cuts it)."""
    world = _size(group)
    if world == 1:
        return t
    fmt = get_memory_format(t)
    rank = dist.get_rank(group=group)
    pieces = [p.contiguous() for p in torch.tensor_split(t, world, dim=-2)]
    lead, n_local, c = tuple(t.shape[:-3]), pieces[rank].shape[-2], t.shape[-1]
    # every rank sends its heads: equal shares, as the reference assumes (it sizes all receive buffers like its own piece)
    recv_shapes = [(*lead, t.shape[-3], n_local, c) for _ in range(world)]
    return torch.cat(_exchange(pieces, recv_shapes, group), dim=-3).contiguous(memory_format=fmt)


# ---------------------------------------------------------------------------------------------------- autograd node
class _Paired(torch.autograd.Function):
    """``forward_fn(x)`` forward, ``backward_fn(grad)`` backward -- both closures over the group and the shard shapes."""

    @staticmethod
    def forward(ctx, x: Tensor, forward_fn: Callable[[Tensor], Tensor], backward_fn: Callable[[Tensor], Tensor]):
        ctx.backward_fn = backward_fn
        return forward_fn(x)

    @staticmethod
    def backward(ctx, grad: Tensor):
        return ctx.backward_fn(grad), None, None


def _paired(x: Tensor, group, forward_fn, backward_fn) -> Tensor:
    if not group:  # no model group: every operator is the identity, forward and backward (as in the reference)
        return x
    return _Paired.apply(x, forward_fn, backward_fn)


# -------------------------------------------------------------------------------------------------------- operators
def shard_tensor(input_: Tensor, dim: int, shapes: tuple, mgroup, gather_in_backward: bool = True) -> Tensor:
    """Keep the part of ``input_`` that belongs to this rank (reference distributed/graph.py:19-44)."""
    return _paired(input_, mgroup, lambda t: _take(t, dim, shapes, mgroup),
                   lambda g: _collect(g, dim, shapes, mgroup, communicate=gather_in_backward))


def gather_tensor(input_: Tensor, dim: int, shapes: tuple, mgroup) -> Tensor:
    """Gather the shards of all ranks along ``dim`` (reference distributed/graph.py:47-68)."""
    return _paired(input_, mgroup, lambda t: _collect(t, dim, shapes, mgroup), lambda g: _take(g, dim, shapes, mgroup))


def reduce_tensor(input_: Tensor, mgroup) -> Tensor:
    """Sum over the ranks; the gradient passes through unchanged (reference distributed/graph.py:71-88)."""
    return _paired(input_, mgroup, lambda t: _total(t, mgroup), lambda g: g)


def sync_tensor(input_: Tensor, dim: int, shapes: tuple, mgroup) -> Tensor:
    """Gather forward; all-reduce, then split, backward (reference distributed/graph.py:91-112)."""
    return _paired(input_, mgroup, lambda t: _collect(t, dim, shapes, mgroup),
                   lambda g: _take(_total(g, mgroup), dim, shapes, mgroup))


def reduce_shard_tensor(input_: Tensor, dim: int, shapes: tuple, mgroup) -> Tensor:
    """All-reduce, then split, forward; gather backward (reference distributed/graph.py:115-137)."""
    return _paired(input_, mgroup, lambda t: _take(_total(t, mgroup), dim, shapes, mgroup),
                   lambda g: _collect(g, dim, shapes, mgroup))


def shard_heads(input_: Tensor, shapes: list, mgroup) -> Tensor:
    """``(batch, ..., heads, sequence shard, channels)`` -> all of the sequence, this rank's heads (reference
    distributed/transformer.py:85-106)."""
    return _paired(input_, mgroup, lambda t: _heads_to_sequence(t, shapes, mgroup),
                   lambda g: _sequence_to_heads(g, shapes, mgroup))


def shard_sequence(input_: Tensor, shapes: list, mgroup) -> Tensor:
    """``(batch, ..., heads shard, sequence, channels)`` -> all heads, this rank's part of the sequence (reference
    distributed/transformer.py:109-130)."""
    return _paired(input_, mgroup, lambda t: _sequence_to_heads(t, shapes, mgroup),
                   lambda g: _heads_to_sequence(g, shapes, mgroup))
