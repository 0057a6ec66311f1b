"""Processor chunks mirroring reference layers/chunk.py: ``nn.ModuleList`` named ``blocks`` run in sequence.

In the reference a chunk is the unit of activation checkpointing; here it only groups blocks so that the
``state_dict`` keys (``proc.<chunk>.blocks.<i>...``) are identical.
"""

from __future__ import annotations

from abc import ABC
from abc import abstractmethod
from typing import Optional

from torch import Tensor
from torch import nn

from ..runtime import EdgePlan
from .block import GraphConvProcessorBlock
from .block import GraphTransformerProcessorBlock
from .block import TransformerProcessorBlock
from .mlp import MLP


class BaseProcessorChunk(nn.Module, ABC):
    def __init__(self, num_channels: int, num_layers: int, *args, activation: str = "GELU", **kwargs) -> None:
        super().__init__()
        self.num_channels = num_channels
        self.num_layers = num_layers

    def build_blocks(self, block, *args, **kwargs) -> None:
        self.blocks = nn.ModuleList([block(*args, **kwargs) for _ in range(self.num_layers)])

    @abstractmethod
    def forward(self, x, shapes, batch_size, model_comm_group=None): ...


class TransformerProcessorChunk(BaseProcessorChunk):
    def __init__(self, num_channels: int, num_layers: int, window_size: int, num_heads: int = 16,
                 mlp_hidden_ratio: int = 4, activation: str = "GELU", dropout_p: float = 0.0) -> None:
        super().__init__(num_channels=num_channels, num_layers=num_layers)
        self.build_blocks(
            TransformerProcessorBlock, num_channels=num_channels, hidden_dim=mlp_hidden_ratio * num_channels,
            num_heads=num_heads, activation=activation, window_size=window_size, dropout_p=dropout_p,
        )

    def native(self, x: Tensor, batch_size: int) -> Tensor:
        for blk in self.blocks:
            x = blk.native(x, batch_size)
        return x

    def forward(self, x: Tensor, shapes: list, batch_size: int, model_comm_group=None):
        for blk in self.blocks:
            x = blk(x, shapes, batch_size, model_comm_group=model_comm_group)
        return (x,)


class GNNProcessorChunk(BaseProcessorChunk):
    def __init__(self, num_channels: int, num_layers: int, mlp_extra_layers: int = 0, activation: str = "SiLU",
                 edge_dim: Optional[int] = None) -> None:
        super().__init__(num_channels=num_channels, num_layers=num_layers)
        if edge_dim:
            self.emb_edges = MLP(in_features=edge_dim, hidden_dim=num_channels, out_features=num_channels,
                                 n_extra_layers=mlp_extra_layers, activation=activation)
        else:
            self.emb_edges = None
        self.build_blocks(GraphConvProcessorBlock, num_channels, num_channels, mlp_extra_layers=mlp_extra_layers,
                          activation=activation)

    def native(self, x: Tensor, e_csr: Tensor, plan: EdgePlan, halo=None):
        """``e_csr``: raw (first chunk) or embedded edge state in CSR order, compute dtype."""
        if self.emb_edges is not None:
            e_csr = self.emb_edges.native()(e_csr)
        for blk in self.blocks:
            x, e_csr = blk.native(x, e_csr, plan, halo)
        return x, e_csr

    def forward(self, x, edge_attr, edge_index, shapes, model_comm_group=None, size=None):
        if self.emb_edges is not None:
            edge_attr = self.emb_edges(edge_attr)
        for blk in self.blocks:
            x, edge_attr = blk(x, edge_attr, edge_index, shapes, model_comm_group, size=size)
        return x, edge_attr


class GraphTransformerProcessorChunk(BaseProcessorChunk):
    def __init__(self, num_channels: int, num_layers: int, num_heads: int = 16, mlp_hidden_ratio: int = 4,
                 activation: str = "GELU", edge_dim: Optional[int] = None) -> None:
        super().__init__(num_channels=num_channels, num_layers=num_layers)
        self.build_blocks(
            GraphTransformerProcessorBlock, in_channels=num_channels, hidden_dim=mlp_hidden_ratio * num_channels,
            out_channels=num_channels, num_heads=num_heads, edge_dim=edge_dim, activation=activation,
        )

    def native(self, x: Tensor, edge_attr_csr: Tensor, plan: EdgePlan, halo=None) -> Tensor:
        for blk in self.blocks:
            x = blk.native(x, edge_attr_csr, plan, halo)
        return x

    def forward(self, x, edge_attr, edge_index, shapes, batch_size, model_comm_group=None, size=None):
        for blk in self.blocks:
            x, edge_attr = blk(x, edge_attr, edge_index, shapes, batch_size, model_comm_group, size=size)
        return x, edge_attr
