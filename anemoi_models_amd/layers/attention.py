"""Multi-head self attention over the mesh nodes, mirroring reference layers/attention.py:34-112.

Parameters: ``lin_qkv`` (no bias by default) and ``projection`` (bias) -- identical ``state_dict`` keys.
"""

from __future__ import annotations

from typing import Optional

from torch import Tensor
from torch import nn


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

    def forward(self, x: Tensor, shapes: list, batch_size: int, model_comm_group=None) -> Tensor:
        raise NotImplementedError("MultiHeadSelfAttention: MI355X kernel not available in this build")
