"""Multi-head self attention over the mesh nodes, mirroring reference layers/attention.py:34-112.

Parameters: ``lin_qkv`` (no bias by default) and ``projection`` (bias) -- identical ``state_dict`` keys.
"""

from __future__ import annotations

import os
from typing import Optional

import torch
from torch import Tensor
from torch import nn

from .. import ops
from .. import runtime
from .mlp import linear_native


class MultiHeadSelfAttention(nn.Module):
    def __init__(self, num_heads: int, embed_dim: int, bias: bool = False, is_causal: bool = False,
                 window_size: Optional[int] = None, dropout_p: float = 0.0):
        super().__init__()
        assert (
            embed_dim % num_heads == 0
        ), f"Embedding dimension ({embed_dim}) must be divisible by number of heads ({num_heads})"
        self.num_heads = num_heads
        self.embed_dim = embed_dim
        self.head_dim = embed_dim // num_heads
        self.window_size = (window_size, window_size)
        self.dropout_p = dropout_p
        self.is_causal = is_causal
        self.lin_qkv = nn.Linear(embed_dim, 3 * embed_dim, bias=bias)
        self.projection = nn.Linear(embed_dim, embed_dim, bias=True)

        self._packed = runtime.PackedWeights()

    def attention_window(self) -> int:
        """-1 (global) by default = the reference's scaled_dot_product_attention fallback, which ignores the window
        (reference layers/attention.py:99-105); ``ANEMOI_AMD_FLASH_WINDOW=1`` applies flash-attn's
        ``window_size=(w, w)`` semantics of the reference's GPU path (layers/attention.py:96)."""
        if os.environ.get("ANEMOI_AMD_FLASH_WINDOW", "0") == "1" and self.window_size[0] is not None:
            return int(self.window_size[0])
        return -1

    def dropout(self, model_comm_group=None):
        """``(dropout_p, seed, seed_dev)`` of this call: the reference drops attention probabilities in training mode only
        (layers/attention.py:90).  Eagerly the seed is drawn per call from torch's CPU generator (``torch.manual_seed``
        reproduces the mask) and ``seed_dev`` is ``None``; inside a ``runtime.DeviceDropout`` context the seed is a
        per-module constant (drawn once, the same way) and the step's device word goes with it -- capturable in a HIP graph.
        With a model group every rank must use ONE seed: rank 0's draw is broadcast -- per call eagerly, once per module
        under ``DeviceDropout`` (the step counters of the ranks advance in lockstep)."""
        if not self.training or self.dropout_p <= 0.0:
            return 0.0, 0, None
        dd = runtime.device_dropout()
        if dd is not None and self.__dict__.get("_layer_seed") is not None:
            return float(self.dropout_p), self.__dict__["_layer_seed"], dd.word
        seed = int(torch.randint(0, 2**31 - 1, (1,)).item())
        if model_comm_group is not None and model_comm_group.size() > 1 and not hasattr(model_comm_group, "rank_"):
            import torch.distributed as dist  # (``rank_``: partition.SimulatedRank, the one-rank-alone timing stand-in)

            seed_t = torch.tensor([seed], dtype=torch.int64)
            if dist.get_backend(model_comm_group) == "nccl":
                seed_t = seed_t.to(self.lin_qkv.weight.device)
            dist.broadcast(seed_t, dist.get_global_rank(model_comm_group, 0), group=model_comm_group)
            seed = int(seed_t.item())
        if dd is None:
            return float(self.dropout_p), seed, None
        self.__dict__["_layer_seed"] = seed
        return float(self.dropout_p), seed, dd.word

    def native(self, x: Tensor, batch_size: int) -> Tensor:
        qkv = linear_native(self._packed, "lin_qkv", self.lin_qkv, x)  # [B*S, 3C] = q | k | v
        p, seed, seed_dev = self.dropout()
        att = ops.mhsa(qkv, batch_size, self.num_heads, self.attention_window(), dropout_p=p, dropout_seed=seed,
                       seed_dev=seed_dev)
        return linear_native(self._packed, "projection", self.projection, att)

    def _sharded(self, x: Tensor, shapes: list, model_comm_group) -> Tensor:
        """The reference's sequence-sharded call (layers/attention.py:72-112 with a model group): every rank holds a row
        range of the sequence; q, k, v go through ``shard_heads`` (all of the sequence, this rank's heads), the attention
        runs on the local heads, ``shard_sequence`` brings the rows back.  The collectives are autograd nodes
        (``distributed/collectives.py``), so the same code trains.  (The model root does not come through here: its
        node-partitioned forward exchanges heads itself, ``distributed/partition.py::HeadExchange``.)"""
        from .. import autograd, training
        from ..distributed.transformer import shard_heads, shard_sequence

        grad = training.wants_grad(self, x)
        dtype = runtime.compute_dtype(x)
        xin = training._cast(x, dtype)
        if grad:
            qkv = autograd.linear(xin, self.lin_qkv.weight, self.lin_qkv.bias)
        else:
            qkv = linear_native(self._packed, "lin_qkv", self.lin_qkv, xin)
        n_local, h, d = qkv.shape[0], self.num_heads, self.head_dim
        # [n_local, 3, H, D] -> three (1, H, n_local, D) tensors, the layout the reference hands to shard_heads
        q, k, v = (qkv.view(n_local, 3, h, d)[:, i].permute(1, 0, 2).unsqueeze(0) for i in range(3))
        q, k, v = (shard_heads(t, shapes=shapes, mgroup=model_comm_group) for t in (q, k, v))
        h_loc, n_all = q.shape[1], q.shape[2]
        fused = torch.stack([t[0].permute(1, 0, 2) for t in (q, k, v)], dim=1).reshape(n_all, 3 * h_loc * d).contiguous()
        # attention dropout across the group: ONE seed (rank 0's draw, broadcast) and the global head index in the mask's
        # hash, so the ranks together drop exactly what the unsharded attention would with that seed
        p, seed, seed_dev = self.dropout(model_comm_group)
        h0 = 0
        if p > 0.0:
            from ..distributed.shapes import split_bounds

            h0 = split_bounds(h, model_comm_group.size())[model_comm_group.rank()]
        if grad:
            att = autograd.mhsa(fused, 1, h_loc, self.attention_window(), p, seed, h0, h, seed_dev)
        else:
            att = ops.mhsa(fused, 1, h_loc, self.attention_window(), dropout_p=p, dropout_seed=seed, head_offset=h0,
                           heads_total=h, seed_dev=seed_dev)
        att = att.view(n_all, h_loc, d).permute(1, 0, 2).unsqueeze(0)  # (1, H_local, N, D)
        att = shard_sequence(att, shapes=shapes, mgroup=model_comm_group)  # (1, H, n_local, D)
        att = att[0].permute(1, 0, 2).reshape(n_local, h * d).contiguous()
        if grad:
            return autograd.linear(att, self.projection.weight, self.projection.bias)
        return linear_native(self._packed, "projection", self.projection, att)

    def forward(self, x: Tensor, shapes: list, batch_size: int, model_comm_group=None) -> Tensor:
        if model_comm_group is not None and model_comm_group.size() > 1:
            assert batch_size == 1, "Only batch size of 1 is supported when model is sharded accross GPUs"
            return self._sharded(x, shapes, model_comm_group)
        from .. import autograd, training

        if training.wants_grad(self, x):  # reference layers/attention.py:67-112 with an autograd graph
            xin = training._cast(x, runtime.compute_dtype(x))
            qkv = autograd.linear(xin, self.lin_qkv.weight, self.lin_qkv.bias)
            p, seed, seed_dev = self.dropout()
            att = autograd.mhsa(qkv, batch_size, self.num_heads, self.attention_window(), p, seed, seed_dev=seed_dev)
            return autograd.linear(att, self.projection.weight, self.projection.bias)
        dtype = runtime.compute_dtype(x)
        xin = x if x.dtype == dtype else x.to(dtype)
        return self.native(xin if xin.stride(-1) == 1 else xin.contiguous(), batch_size)
