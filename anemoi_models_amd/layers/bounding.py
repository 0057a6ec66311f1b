"""Output boundings mirroring reference layers/bounding.py:60-124 (in-place clamps on selected output variables).

Two routes, same arithmetic:

* ``module(x)`` -- the reference's own in-place index assignments in plain torch (any device); what a user calling a
  bounding module directly gets, and what the reference's exact-value tests exercise;
* inside the model (``AnemoiModelEncProcDec._finish``) the whole ``boundings`` list is compiled ONCE into an ordered op
  list (:func:`compile_boundings`) and applied by one kernel, ``anemoi_bound_output`` -- one pass over the touched
  columns instead of 3-5 indexing kernels per module, no host-resident index tensors on the hot path.
"""

from __future__ import annotations

import math
from abc import ABC
from abc import abstractmethod
from typing import List, Optional, Sequence, Tuple

import torch
from torch import nn


def _indices(variables, name_to_index) -> torch.Tensor:
    """Sorted output columns of ``variables`` (reference data_indices/tensor.py:91-94 ``_build_idx_from_includes``)."""
    missing = [v for v in variables if v not in name_to_index]
    assert not missing, f"Data indexing has invalid entries {missing}, not in dataset."
    return torch.tensor(sorted(name_to_index[v] for v in variables), dtype=torch.int64)


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


# one op = (column, lo, hi, multiplier column or -1); see include/anemoi_amd.h: anemoi_bound_output
Op = Tuple[int, float, float, int]


def compile_boundings(boundings: Sequence[nn.Module]) -> Optional[List[Op]]:
    """The ordered op list equivalent to calling ``boundings`` one after the other, or ``None`` when the list holds a
    class this module does not know (a user subclass: the model then calls the modules themselves)."""
    ops: List[Op] = []
    for b in boundings:
        cols = sorted(set(int(c) for c in b.data_index))
        if type(b) is ReluBounding:
            ops += [(c, 0.0, math.inf, -1) for c in cols]
        elif type(b) is HardtanhBounding:
            ops += [(c, float(b.min_val), float(b.max_val), -1) for c in cols]
        elif type(b) is FractionBounding:
            total = int(b.total_variable[0])
            ops += [(c, float(b.min_val), float(b.max_val), -1) for c in cols]
            # ``x[..., idx] *= x[..., total]`` reads the total BEFORE any write: if the total is itself bounded by this
            # module its own product goes last
            ops += [(c, -math.inf, math.inf, total) for c in cols if c != total]
            ops += [(c, -math.inf, math.inf, total) for c in cols if c == total]
        else:
            return None
    return ops


def bounded_columns(ops: Sequence[Op]) -> List[int]:
    """Columns an op list writes or reads: they must stay normalised until the boundings have run."""
    return sorted({c for c, _, _, _ in ops} | {m for _, _, _, m in ops if m >= 0})
