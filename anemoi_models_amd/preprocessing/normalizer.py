"""``InputNormalizer`` mirroring reference preprocessing/normalizer.py:23-205: per-variable affine map
``x * _norm_mul + _norm_add`` with the methods mean-std / std / min-max / max / none, optional statistics remapping,
the same persistent buffers (``_norm_mul``, ``_norm_add``, ``_input_idx``, ``_output_idx``) and the same shape-driven
choice of the index set in ``transform`` / ``inverse_transform``."""

from __future__ import annotations

import logging
import warnings
from typing import Optional

import numpy as np
import torch

from . import BasePreprocessor

LOGGER = logging.getLogger(__name__)

METHODS = ("mean-std", "std", "min-max", "max", "none")


class InputNormalizer(BasePreprocessor):
    """Normalizes input data with a configurable method."""

    def __init__(self, config=None, data_indices=None, statistics: Optional[dict] = None) -> None:
        super().__init__(config, data_indices, statistics)
        name_to_index = self.data_indices.data.input.name_to_index
        minimum, maximum = np.array(statistics["minimum"], copy=True), np.array(statistics["maximum"], copy=True)
        mean, stdev = np.array(statistics["mean"], copy=True), np.array(statistics["stdev"], copy=True)

        # optionally reuse the statistics of one variable for another one (two steps: order independent)
        remapped = {}
        for remap, source in self.remap.items():
            i_src, i_dst = name_to_index[source], name_to_index[remap]
            remapped[i_dst] = (minimum[i_src], maximum[i_src], mean[i_src], stdev[i_src])
        for idx, new in remapped.items():
            minimum[idx], maximum[idx], mean[idx], stdev[idx] = new

        self._validate_normalization_inputs(name_to_index, minimum, maximum, mean, stdev)
        norm_add = np.zeros((minimum.size,), dtype=np.float32)
        norm_mul = np.ones((minimum.size,), dtype=np.float32)
        for name, i in name_to_index.items():
            method = self.methods.get(name, self.default)
            if method == "mean-std":
                if stdev[i] < (mean[i] * 1e-6):
                    warnings.warn(f"Normalizing: the field seems to have only one value {mean[i]}")
                norm_mul[i] = 1 / stdev[i]
                norm_add[i] = -mean[i] / stdev[i]
            elif method == "std":
                if stdev[i] < (mean[i] * 1e-6):
                    warnings.warn(f"Normalizing: the field seems to have only one value {mean[i]}")
                norm_mul[i] = 1 / stdev[i]
                norm_add[i] = 0
            elif method == "min-max":
                span = maximum[i] - minimum[i]
                if span < 1e-9:
                    warnings.warn(f"Normalizing: the field {name} seems to have only one value {maximum[i]}.")
                norm_mul[i] = 1 / span
                norm_add[i] = -minimum[i] / span
            elif method == "max":
                norm_mul[i] = 1 / maximum[i]
            elif method == "none":
                LOGGER.info("Normalizing: %s is not normalized.", name)
            else:
                raise ValueError(f"Unknown normalisation method for {name}: {method}")

        self.register_buffer("_norm_mul", torch.from_numpy(norm_mul), persistent=True)
        self.register_buffer("_norm_add", torch.from_numpy(norm_add), persistent=True)
        self.register_buffer("_input_idx", torch.as_tensor(data_indices.data.input.full), persistent=True)
        self.register_buffer("_output_idx", torch.as_tensor(data_indices.data.output.full), persistent=True)

    def _validate_normalization_inputs(self, name_to_index: dict, minimum, maximum, mean, stdev) -> None:
        assert len(self.methods) == sum(len(v) for v in self.method_config.values()), (
            f"Error parsing methods in InputNormalizer methods ({len(self.methods)}) and entries in config do not match.")
        n = minimum.size
        assert maximum.size == n and mean.size == n and stdev.size == n, (maximum.size, mean.size, stdev.size, n)
        for name, method in self.methods.items():
            assert name in name_to_index, f"{name} is not a valid variable name"
            assert method in METHODS, f"{method} is not a valid normalisation method"

    def _affine(self, x: torch.Tensor, data_index: Optional[torch.Tensor], which: str):
        if data_index is not None:
            idx = data_index
        elif x.shape[-1] == len(getattr(self, which)):
            idx = getattr(self, which)
        else:
            return self._norm_mul, self._norm_add
        idx = idx.long()
        return self._norm_mul[idx], self._norm_add[idx]

    def transform(self, x: torch.Tensor, in_place: bool = True, data_index: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``x * mul + add`` on ``[..., nvars]``: the full variable list, the input variables or ``data_index``."""
        if not in_place:
            x = x.clone()
        mul, add = self._affine(x, data_index, "_input_idx")
        x[..., :] = x[..., :] * mul + add
        return x

    def inverse_transform(self, x: torch.Tensor, in_place: bool = True,
                          data_index: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``(x - add) / mul`` on ``[..., nvars | nvars_pred]``."""
        if not in_place:
            x = x.clone()
        mul, add = self._affine(x, data_index, "_output_idx")
        x[..., :] = (x[..., :] - add) / mul
        return x
