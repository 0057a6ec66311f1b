"""Pre- / post-processors mirroring reference preprocessing/__init__.py:20-194 (same class names, config schema,
``forward(x, in_place, inverse)`` contract).  Only the pieces on the inference path are provided: the base class, the
``Processors`` container, ``normalizer.InputNormalizer`` and the NaN imputers of ``imputer``; the remappers are not
part of this build."""

from __future__ import annotations

import logging
from typing import Optional

import torch
from torch import Tensor
from torch import nn

LOGGER = logging.getLogger(__name__)


class BasePreprocessor(nn.Module):
    """Base class for data pre- and post-processors (reference preprocessing/__init__.py:20-132)."""

    def __init__(self, config=None, data_indices=None, statistics: Optional[dict] = None) -> None:
        super().__init__()
        self.default, self.remap, self.method_config = self._process_config(config)
        self.methods = self._invert_key_value_list(self.method_config)
        self.data_indices = data_indices

    @classmethod
    def _process_config(cls, config):
        special = ["default", "remap"]
        default = config.get("default", "none")
        remap = config.get("remap", {})
        method_config = {k: v for k, v in config.items() if k not in special and v is not None and v != "none"}
        if not method_config:
            LOGGER.warning("%s: Using default method %s for all variables not specified in the config.", cls.__name__,
                           default)
        for m in method_config:
            if isinstance(method_config[m], str):
                method_config[m] = {method_config[m]: f"{m}_{method_config[m]}"}
            elif isinstance(method_config[m], list):
                method_config[m] = {method: f"{m}_{method}" for method in method_config[m]}
        return default, remap, method_config

    @staticmethod
    def _invert_key_value_list(method_config: dict) -> dict:
        return {variable: method for method, variables in method_config.items() if not isinstance(variables, str)
                for variable in variables}

    def forward(self, x, in_place: bool = True, inverse: bool = False) -> Tensor:
        if inverse:
            return self.inverse_transform(x, in_place=in_place)
        return self.transform(x, in_place=in_place)

    def transform(self, x, in_place: bool = True) -> Tensor:
        return x if in_place else x.clone()

    def inverse_transform(self, x, in_place: bool = True) -> Tensor:
        return x if in_place else x.clone()


class Processors(nn.Module):
    """An ordered collection of processors; the inverse collection runs them back to front
    (reference preprocessing/__init__.py:135-194)."""

    def __init__(self, processors: list, inverse: bool = False) -> None:
        super().__init__()
        self.inverse = inverse
        self.first_run = True
        if inverse:
            processors = processors[::-1]
        self.processors = nn.ModuleDict(processors)

    def __repr__(self) -> str:
        return f"{self.__class__.__name__} [{'inverse' if self.inverse else 'forward'}]({self.processors})"

    def forward(self, x, in_place: bool = True) -> Tensor:
        for processor in self.processors.values():
            x = processor(x, in_place=in_place, inverse=self.inverse)
        if self.first_run:
            self.first_run = False
            self._run_checks(x)
        return x

    def _run_checks(self, x) -> None:
        if not self.inverse:
            assert not torch.isnan(x).any(), (
                f"NaNs ({torch.isnan(x).sum()}) found in processed tensor after {self.__class__.__name__}.")
