"""Node / edge attribute holders mirroring reference layers/graph.py.

These modules only own parameters and buffers (identical ``state_dict`` keys:
``trainable``, ``latlons_<name>``, ``trainable_tensors.<name>.trainable``).  On the MI355X path the
concatenation they describe is performed by ``anemoi_assemble_nodes`` / ``anemoi_edge_attr_csr``
directly into the kernels' input buffers; the ``forward`` methods below keep the reference
semantics (plain tensor result) for callers that want the attribute matrix itself.
"""

from __future__ import annotations

import torch
from torch import Tensor
from torch import nn


class TrainableTensor(nn.Module):
    """``[tensor_size, trainable_size]`` zero-initialised parameter appended to a fixed attribute matrix.

    Reference layers/graph.py:18-44.
    """

    def __init__(self, tensor_size: int, trainable_size: int) -> None:
        super().__init__()
        if trainable_size > 0:
            trainable = nn.Parameter(torch.zeros(tensor_size, trainable_size))
        else:
            trainable = None
        self.register_parameter("trainable", trainable)

    def forward(self, x: Tensor, batch_size: int) -> Tensor:
        parts = [x.repeat(batch_size, 1)]
        if self.trainable is not None:
            parts.append(self.trainable.to(x.device).repeat(batch_size, 1))
        return torch.cat(parts, dim=-1)


class NamedNodesAttributes(nn.Module):
    """sin/cos coordinates (persistent buffers) + trainable tensor per node set.  Reference layers/graph.py:47-113."""

    def __init__(self, num_trainable_params: int, graph_data) -> None:
        super().__init__()
        self.define_fixed_attributes(graph_data, num_trainable_params)
        self.trainable_tensors = nn.ModuleDict()
        for name, nodes in graph_data.node_items():
            self.register_coordinates(name, nodes.x)
            self.register_tensor(name, num_trainable_params)

    def define_fixed_attributes(self, graph_data, num_trainable_params: int) -> None:
        """Node counts and attribute widths (2 x coordinate dims + trainable) per node set."""
        names = list(graph_data.node_types)
        self.num_nodes = {n: graph_data[n].num_nodes for n in names}
        self.attr_ndims = {n: 2 * graph_data[n].x.shape[1] + num_trainable_params for n in names}

    def register_coordinates(self, name: str, node_coords: Tensor) -> None:
        """``[sin(coords) | cos(coords)]`` as the persistent buffer ``latlons_<name>``."""
        self.register_buffer(f"latlons_{name}", torch.cat([torch.sin(node_coords), torch.cos(node_coords)], dim=-1),
                             persistent=True)

    def register_tensor(self, name: str, num_trainable_params: int) -> None:
        self.trainable_tensors[name] = TrainableTensor(self.num_nodes[name], num_trainable_params)

    def get_coordinates(self, name: str) -> Tensor:
        sc = getattr(self, f"latlons_{name}")
        half = sc.shape[1] // 2
        return torch.atan2(sc[:, :half], sc[:, half:])

    def latlons(self, name: str) -> Tensor:
        return getattr(self, f"latlons_{name}")

    def forward(self, name: str, batch_size: int) -> Tensor:
        return self.trainable_tensors[name](self.latlons(name), batch_size)
