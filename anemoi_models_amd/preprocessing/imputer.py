"""NaN imputers mirroring reference preprocessing/imputer.py:24-305 (same class names, config schema, attribute names
``nan_locations`` / ``loss_mask_training`` / ``index_*`` / ``replacement`` and ``transform`` / ``inverse_transform``
contract), written as ONE masked select over the imputed columns instead of a Python loop of boolean-index assignments
per variable (on the device that loop is a ``nonzero`` + scatter + host sync per variable and call).

* ``InputImputer``: replacement = a statistic of the variable (``mean``, ``minimum`` ...), config ``{statistic: [vars]}``.
* ``ConstantImputer``: replacement = the config key itself, config ``{value: [vars]}``.
* The NaN map is taken ONCE, from the first tensor seen (first element of every leading dimension, reference :110-114),
  and re-used: later calls overwrite exactly those grid points, and ``inverse_transform`` puts NaN back there.
* ``Dynamic*``: the map is recomputed per call and the inverse is the identity (reference :234-273).
"""

from __future__ import annotations

import warnings
from typing import List
from typing import Optional

import torch
from torch import Tensor

from . import BasePreprocessor


class BaseImputer(BasePreprocessor):
    """Shared bookkeeping: which dataset variable is imputed, with what, and where it sits in the four layouts
    (training / inference x input / output)."""

    def __init__(self, config=None, data_indices=None, statistics: Optional[dict] = None) -> None:
        super().__init__(config, data_indices, statistics)
        self.nan_locations = None
        self.loss_mask_training = None  # [grid, n model outputs]: 0 where an imputed output value is not real data

    # ------------------------------------------------------------------ construction
    def _create_imputation_indices(self, statistics=None) -> None:
        """reference :60-104 -- one entry per imputed variable of the training input, in its order."""
        train_in = self.data_indices.data.input.name_to_index
        infer_in = self.data_indices.model.input.name_to_index
        train_out = self.data_indices.data.output.name_to_index
        infer_out = self.data_indices.model.output.name_to_index
        self.num_training_input_vars, self.num_inference_input_vars = len(train_in), len(infer_in)
        self.num_training_output_vars, self.num_inference_output_vars = len(train_out), len(infer_out)
        self.index_training_input: List[int] = []
        self.index_inference_input: List[Optional[int]] = []
        self.index_training_output: List[Optional[int]] = []
        self.index_inference_output: List[Optional[int]] = []
        self.replacement: list = []
        if statistics is not None and not isinstance(statistics, dict):
            raise TypeError(f"Statistics {type(statistics)} is optional and not a dictionary")
        for name, position in train_in.items():
            method = self.methods.get(name, self.default)
            if method == "none":
                continue
            self.index_training_input.append(position)
            self.index_training_output.append(train_out.get(name))
            self.index_inference_input.append(infer_in.get(name))
            self.index_inference_output.append(infer_out.get(name))
            if statistics is None:
                self.replacement.append(method)  # the config key is the value
            else:
                assert method in statistics, f"{method} is not a method in the statistics metadata"
                self.replacement.append(statistics[method][position])

    def _validate_indices(self) -> None:
        n = len(self.replacement)
        assert len(self.index_training_input) == len(self.index_inference_input) <= n, (
            f"Error creating imputation indices {len(self.index_training_input)}, "
            f"{len(self.index_inference_input)}, {n}")
        assert len(self.index_training_output) == len(self.index_inference_output) <= n, (
            f"Error creating imputation indices {len(self.index_training_output)}, "
            f"{len(self.index_inference_output)}, {n}")

    # ------------------------------------------------------------------ helpers
    def _layout(self, width: int, train_n: int, infer_n: int, train_idx, infer_idx, what: str):
        if width == train_n:
            return train_idx
        if width == infer_n:
            return infer_idx
        raise ValueError(f"{what} tensor ({width}) does not match the training ({train_n}) or inference shape "
                         f"({infer_n})")

    def _columns(self, layout, device):
        """(source columns in the NaN map, destination columns in x, replacement values) of the variables present in
        this layout, as index tensors."""
        keep = [i for i, dst in enumerate(layout) if dst is not None]
        src = torch.tensor([self.index_training_input[i] for i in keep], dtype=torch.long, device=device)
        dst = torch.tensor([layout[i] for i in keep], dtype=torch.long, device=device)
        val = torch.tensor([float(self.replacement[i]) for i in keep], dtype=torch.float32, device=device)
        return src, dst, val

    def get_nans(self, x: Tensor) -> Tensor:
        """``[grid, variables]`` NaN map of the first element of every leading dimension."""
        return torch.isnan(x.reshape(-1, x.shape[-2], x.shape[-1])[0])

    def _expand_subset_mask(self, x: Tensor, idx_src: int) -> Tensor:
        """NaN map of one source variable broadcast over the leading dimensions of ``x`` (reference :106-108); the
        transforms use the batched ``_fill`` over all variables of a layout instead of one masked store per variable."""
        return self.nan_locations[:, idx_src].expand(*x.shape[:-2], -1)

    def _fill(self, x: Tensor, mask: Tensor, dst: Tensor, val) -> Tensor:
        if dst.numel() > 0:
            cols = x[..., dst]
            x[..., dst] = torch.where(mask, val.to(cols.dtype) if isinstance(val, Tensor) else val, cols)
        return x

    # ------------------------------------------------------------------ transform / inverse
    def transform(self, x: Tensor, in_place: bool = True) -> Tensor:
        if not in_place:
            x = x.clone()
        if self.nan_locations is None:
            self.nan_locations = self.get_nans(x)
            n_out = len(self.data_indices.model.output.name_to_index)
            self.loss_mask_training = torch.ones((x.shape[-2], n_out), device=x.device)
            for src, dst in zip(self.index_training_input, self.index_inference_output):
                if dst is not None:
                    self.loss_mask_training[:, dst] = (~self.nan_locations[:, src]).to(self.loss_mask_training.dtype)
        layout = self._layout(x.shape[-1], self.num_training_input_vars, self.num_inference_input_vars,
                              self.index_training_input, self.index_inference_input, "Input")
        src, dst, val = self._columns(layout, x.device)
        return self._fill(x, self.nan_locations.to(x.device)[:, src], dst, val)

    def inverse_transform(self, x: Tensor, in_place: bool = True) -> Tensor:
        if not in_place:
            x = x.clone()
        layout = self._layout(x.shape[-1], self.num_training_output_vars, self.num_inference_output_vars,
                              self.index_training_output, self.index_inference_output, "Input")
        src, dst, _ = self._columns(layout, x.device)
        return self._fill(x, self.nan_locations.to(x.device)[:, src], dst, float("nan"))


class InputImputer(BaseImputer):
    """Imputes NaNs with a statistic of the variable (reference :176-202)."""

    def __init__(self, config=None, data_indices=None, statistics: Optional[dict] = None) -> None:
        super().__init__(config, data_indices, statistics)
        self._create_imputation_indices(statistics)
        self._validate_indices()


class ConstantImputer(BaseImputer):
    """Imputes NaNs with the constant given as the config key (reference :205-231)."""

    def __init__(self, config=None, data_indices=None, statistics: Optional[dict] = None) -> None:
        super().__init__(config, data_indices, statistics)
        self._create_imputation_indices()
        self._validate_indices()


class DynamicMixin:
    """NaN map recomputed on every call; the inverse leaves the imputed values in place (reference :234-273)."""

    def get_nans(self, x: Tensor) -> Tensor:
        return torch.isnan(x)

    def transform(self, x: Tensor, in_place: bool = True) -> Tensor:
        if not in_place:
            x = x.clone()
        nan_locations = self.get_nans(x)
        self.loss_mask_training = torch.ones((x.shape[-2], len(self.data_indices.model.output.name_to_index)),
                                             device=x.device)
        layout = self._layout(x.shape[-1], self.num_training_input_vars, self.num_inference_input_vars,
                              self.index_training_input, self.index_inference_input, "Input")
        src, dst, val = self._columns(layout, x.device)
        return self._fill(x, nan_locations[..., src], dst, val)

    def inverse_transform(self, x: Tensor, in_place: bool = True) -> Tensor:
        return x


_DYNAMIC_WARNING = ("You are using a dynamic Imputer: NaN values will not be present in the model predictions. "
                    "The model will be trained to predict imputed values. This might deteriorate performances.")


class DynamicInputImputer(DynamicMixin, InputImputer):
    def __init__(self, config=None, data_indices=None, statistics: Optional[dict] = None) -> None:
        super().__init__(config, data_indices, statistics)
        warnings.warn(_DYNAMIC_WARNING)


class DynamicConstantImputer(DynamicMixin, ConstantImputer):
    def __init__(self, config=None, data_indices=None, statistics: Optional[dict] = None) -> None:
        super().__init__(config, data_indices, statistics)
        warnings.warn(_DYNAMIC_WARNING)
