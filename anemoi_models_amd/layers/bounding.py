"""Output boundings mirroring reference layers/bounding.py:60-124 (in-place clamps on selected output variables).

Trivial, optional and elementwise on a handful of columns: plain torch indexing on the device tensor.
"""

from __future__ import annotations

from abc import ABC
from abc import abstractmethod

import torch
from torch import nn


def _indices(variables, name_to_index) -> torch.Tensor:
    return torch.tensor([name_to_index[v] for v in variables if v in name_to_index], dtype=torch.int64)


class BaseBounding(nn.Module, ABC):
    def __init__(self, *, variables: list, name_to_index: dict) -> None:
        super().__init__()
        self.name_to_index = name_to_index
        self.variables = variables
        self.data_index = self._create_index(variables)

    def _create_index(self, variables: list) -> torch.Tensor:
        return _indices(variables, self.name_to_index)

    @abstractmethod
    def forward(self, x: torch.Tensor) -> torch.Tensor: ...


class ReluBounding(BaseBounding):
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x[..., self.data_index] = torch.nn.functional.relu(x[..., self.data_index])
        return x


class HardtanhBounding(BaseBounding):
    def __init__(self, *, variables: list, name_to_index: dict, min_val: float, max_val: float) -> None:
        super().__init__(variables=variables, name_to_index=name_to_index)
        self.min_val = min_val
        self.max_val = max_val

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x[..., self.data_index] = torch.nn.functional.hardtanh(x[..., self.data_index], min_val=self.min_val,
                                                               max_val=self.max_val)
        return x


class FractionBounding(HardtanhBounding):
    def __init__(self, *, variables: list, name_to_index: dict, min_val: float, max_val: float,
                 total_var: str) -> None:
        super().__init__(variables=variables, name_to_index=name_to_index, min_val=min_val, max_val=max_val)
        self.total_variable = self._create_index([total_var])

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = super().forward(x)
        x[..., self.data_index] *= x[..., self.total_variable]
        return x
