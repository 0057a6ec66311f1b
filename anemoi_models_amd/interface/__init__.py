"""``AnemoiModelInterface`` mirroring reference interface/__init__.py:20-123: the wrapper anemoi-inference /
anemoi-training hold -- pre-processors, the model, post-processors, ``predict_step``.  Same constructor kwargs, same
sub-module names (``pre_processors``, ``post_processors``, ``model``) and therefore the same ``state_dict``."""

from __future__ import annotations

import uuid

import torch

from ..models.encoder_processor_decoder import instantiate
from ..preprocessing import Processors


class AnemoiModelInterface(torch.nn.Module):
    def __init__(self, *, config, graph_data, statistics: dict, data_indices, metadata: dict,
                 supporting_arrays: dict = None) -> None:
        super().__init__()
        self.config = config
        self.id = str(uuid.uuid4())
        self.multi_step = self.config.training.multistep_input
        self.graph_data = graph_data
        self.statistics = statistics
        self.metadata = metadata
        self.supporting_arrays = supporting_arrays if supporting_arrays is not None else {}
        self.data_indices = data_indices
        self._build_model()

    def _build_model(self) -> None:
        processors = [
            [name, instantiate(processor, data_indices=self.data_indices, statistics=self.statistics)]
            for name, processor in self.config.data.processors.items()
        ]
        self.pre_processors = Processors(processors)
        self.post_processors = Processors(processors, inverse=True)
        self.model = instantiate(self.config.model.model, model_config=self.config, data_indices=self.data_indices,
                                 graph_data=self.graph_data, _recursive_=False)
        self.forward = self.model.forward

    def predict_step(self, batch: torch.Tensor) -> torch.Tensor:
        """``[batch, time, grid, variables]`` (physical values, input variables) -> de-normalised prediction
        ``[batch, ensemble = 1, grid, output variables]`` (reference interface/__init__.py:97-123)."""
        batch = self.pre_processors(batch, in_place=False)
        with torch.no_grad():
            assert len(batch.shape) == 4, (
                f"The input tensor has an incorrect shape: expected a 4-dimensional tensor, got {batch.shape}!")
            x = batch[:, 0 : self.multi_step, None, ...]  # dummy ensemble dimension as 3rd index
            y_hat = self(x)
        return self.post_processors(y_hat, in_place=False)
