"""Shard-shape helpers with the reference's semantics (reference distributed/shapes.py:19-29)."""

from __future__ import annotations

import torch
from torch import Tensor


def _group_size(group) -> int:
    if group is None:
        return 1
    import torch.distributed as dist

    return dist.get_world_size(group=group)


def get_shape_shards(tensor: Tensor, dim: int, model_comm_group=None) -> list:
    """Per-rank ``list(shape)`` of ``torch.tensor_split(tensor, world, dim)``: the first ``N % P`` shards get one extra row."""
    assert dim < tensor.dim(), f"Error, tensor dimension is {tensor.dim()} which cannot be split along {dim}"
    return [list(x.shape) for x in torch.tensor_split(tensor, _group_size(model_comm_group), dim=dim)]


def change_channels_in_shape(shape_list: list, channels: int) -> list:
    return [x[:-1] + [channels] for x in shape_list] if shape_list else []


def split_bounds(n: int, parts: int) -> list:
    """Row bounds ``[b_0 = 0, ..., b_parts = n]`` of ``tensor_split`` (same node -> rank map as the reference)."""
    base, extra = divmod(n, parts)
    bounds = [0]
    for r in range(parts):
        bounds.append(bounds[-1] + base + (1 if r < extra else 0))
    return bounds
