"""``AnemoiModelInterface`` mirroring reference interface/__init__.py:20-123: the wrapper anemoi-inference /
anemoi-training hold -- pre-processors, the model, post-processors, ``predict_step``.  Same constructor kwargs, same
sub-module names (``pre_processors``, ``post_processors``, ``model``) and therefore the same ``state_dict``."""

from __future__ import annotations

import uuid

import torch

from .. import ops
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
        fused = self._normalizer_affines(batch)
        if fused is not None:
            # the only processor is an InputNormalizer: its per-variable affine maps ride on the first and the last
            # kernel of the forward (anemoi_assemble_nodes / anemoi_finalize_output) -- two passes over the grid state
            # and two over the prediction less, same arithmetic
            with torch.no_grad():
                assert len(batch.shape) == 4, (
                    f"The input tensor has an incorrect shape: expected a 4-dimensional tensor, got {batch.shape}!")
                x = batch[:, 0 : self.multi_step, None, ...]
                return self.model(x, input_affine=fused[0], output_affine=fused[1])
        batch = self.pre_processors(batch, in_place=False)
        with torch.no_grad():
            assert len(batch.shape) == 4, (
                f"The input tensor has an incorrect shape: expected a 4-dimensional tensor, got {batch.shape}!")
            x = batch[:, 0 : self.multi_step, None, ...]  # dummy ensemble dimension as 3rd index
            y_hat = self(x)
        return self.post_processors(y_hat, in_place=False)

    def _normalizer_affines(self, batch: torch.Tensor):
        """``((mul_in, add_in), (mul_out, add_out))`` when pre / post-processing is exactly one ``InputNormalizer``
        acting on the model's input / output variable lists and the model accepts them; else ``None``.
        ``ANEMOI_AMD_FUSE_NORMALIZER=0`` turns the fusion off (A/B)."""
        import inspect
        import os

        from ..preprocessing.normalizer import InputNormalizer

        mode = os.environ.get("ANEMOI_AMD_FUSE_NORMALIZER", "1")
        if mode == "0" or batch.dim() != 4 or not (batch.is_cuda or mode == "force"):
            return None
        procs = list(self.pre_processors.processors.values())
        if len(procs) != 1 or type(procs[0]) is not InputNormalizer:
            return None
        if "input_affine" not in inspect.signature(self.model.forward).parameters:
            return None
        norm = procs[0]
        if batch.shape[-1] != norm._input_idx.numel():  # the full-variable layout is left to the generic route
            return None
        if self.pre_processors.first_run:  # keep the reference's one-off NaN check of the first processed batch
            return None
        i_in, i_out = norm._input_idx.long(), norm._output_idx.long()
        return ((norm._norm_mul[i_in], norm._norm_add[i_in]), (norm._norm_mul[i_out], norm._norm_add[i_out]))

    # ------------------------------------------------------------------ autoregressive rollout (BASELINE config 4)
    def _advance_map(self, device) -> torch.Tensor:
        """int32 ``[V_in]`` column map of ``anemoi_advance_input``: prognostic inputs <- their output column, forcing
        inputs <- their position in the forcing tensor (-2 - k), everything else persists (-1)."""
        idx = self.data_indices.internal_model
        cmap = torch.full((len(idx.input),), -1, dtype=torch.int32)
        cmap[idx.input.prognostic.long()] = idx.output.prognostic.to(torch.int32)
        forcing = idx.input.forcing.long()
        cmap[forcing] = -2 - torch.arange(forcing.numel(), dtype=torch.int32)
        return cmap.to(device)

    def rollout(self, batch: torch.Tensor, n_steps: int, forcings: torch.Tensor = None, model_comm_group=None,
                gather: str = "all"):
        """``n_steps`` autoregressive forecasts from ``batch`` ``[batch, time, grid, input variables]`` (physical
        values).  ``forcings`` ``[n_steps, batch, grid, n_forcing]`` holds the physical forcing inputs valid at each
        step's output time (ordered like ``data_indices.internal_model.input.forcing``); without it the last forcing
        values persist.  Returns ``[n_steps, batch, 1, grid, output variables]`` de-normalised predictions.

        Not part of the reference repository (which stops at :meth:`predict_step`): the loop follows its caller,
        anemoi-training's ``advance_input``.  The state stays on the device in normalised model space; the time shift
        and the prognostic / forcing write-back are one in-place HIP kernel (``anemoi_advance_input``).

        With a model communication group, ``gather="last"`` keeps the state SHARDED between the steps: the intermediate
        forecasts are not all-gathered (each rank holds the grid rows it decodes plus the few its encoder reads, fetched
        by one small all-to-all-v per step) and only the last step's full prediction is returned, ``[1, batch, 1, grid,
        output variables]``.  ``gather="all"`` (default) returns every step, each all-gathered as the reference's decoder
        contract prescribes."""
        if gather not in ("all", "last"):
            raise ValueError(f"rollout: gather must be 'all' or 'last', got {gather!r}")
        keep_sharded = gather == "last" and model_comm_group is not None and model_comm_group.size() > 1
        idx = self.data_indices.internal_model
        x = self.pre_processors(batch, in_place=False)
        assert len(x.shape) == 4, f"The input tensor has an incorrect shape: expected 4 dimensions, got {x.shape}!"
        x = x[:, 0 : self.multi_step, None, ...].float().contiguous().clone()
        cmap = self._advance_map(x.device)
        f_idx = idx.input.forcing.long().to(x.device)
        outs = []
        with torch.no_grad():
            for step in range(n_steps):
                last = step + 1 == n_steps
                y_local = None
                if keep_sharded:
                    from ..distributed.partition import advance_sharded_state, sharded_forward, sharded_state_output

                    y_local, shard_plan = sharded_forward(self.model, x, model_comm_group, local_output=True)
                    if last:  # the one all-gather of the rollout
                        y_hat = sharded_state_output(self.model, x, y_local, shard_plan, model_comm_group).to(x.dtype)
                        outs.append(self.post_processors(y_hat, in_place=False))
                else:
                    y_hat = self.model(x, model_comm_group) if model_comm_group is not None else self.model(x)
                    outs.append(self.post_processors(y_hat, in_place=False))
                if last:
                    break
                f_norm = None
                if forcings is not None and f_idx.numel() > 0:
                    # normalise the forcing columns with the same pre-processors: embed them in a full-width slice
                    full = x[:, -1, 0].clone()
                    full[..., f_idx] = forcings[step].to(full)
                    f_norm = self.pre_processors(full[:, None], in_place=False)[:, 0][..., f_idx][:, None].contiguous()
                if y_local is not None:
                    advance_sharded_state(self.model, x, y_local, shard_plan, cmap, f_norm)
                else:
                    ops.advance_input(x, y_hat.float().contiguous(), cmap, f_norm)
        return torch.stack(outs)
