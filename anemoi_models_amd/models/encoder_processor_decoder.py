"""Model root mirroring reference models/encoder_processor_decoder.py:30-233.

Same constructor (``model_config``, ``data_indices``, ``graph_data``), same sub-module names (``node_attributes``,
``encoder``, ``processor``, ``decoder``, ``boundings``) and therefore the same ``state_dict``; ``forward(x,
model_comm_group=None)`` maps ``[B, T, Ens, G, V_in]`` to ``[B, Ens, G, V_out]``.

The forward is one stream of HIP launches: input assembly -> encoder -> processor (+ skip) -> decoder (f32
output from the last GEMM epilogue) -> prognostic residual -> boundings.  Activation checkpointing of the
reference (``_run_mapper``) has no forward-pass effect and is not needed here.
"""

from __future__ import annotations

import logging
from typing import Optional

import torch
from torch import Tensor
from torch import nn

from .. import ops
from .. import runtime
from ..layers.graph import NamedNodesAttributes

try:  # real hydra when available, otherwise the local minimal implementation
    from hydra.utils import instantiate as _hydra_instantiate  # type: ignore
except ImportError:  # pragma: no cover - hydra is not in the build image
    _hydra_instantiate = None

from ..utils.config import instantiate as _local_instantiate
from ..utils.config import REFERENCE_PREFIX

LOGGER = logging.getLogger(__name__)


def instantiate(cfg, **kwargs):
    target = cfg.get("_target_", "") if hasattr(cfg, "get") else ""
    if _hydra_instantiate is not None and not target.startswith(REFERENCE_PREFIX):
        return _hydra_instantiate(cfg, **kwargs)
    return _local_instantiate(cfg, **kwargs)


class AnemoiModelEncProcDec(nn.Module):
    """Encoder - processor - decoder graph network on MI355X kernels."""

    def __init__(self, *, model_config, data_indices, graph_data) -> None:
        super().__init__()
        self._graph_data = graph_data
        self._graph_name_data = model_config.graph.data
        self._graph_name_hidden = model_config.graph.hidden

        self._calculate_shapes_and_indices(data_indices)
        self._assert_matching_indices(data_indices)
        self.data_indices = data_indices

        self.multi_step = model_config.training.multistep_input
        self.num_channels = model_config.model.num_channels

        self.node_attributes = NamedNodesAttributes(model_config.model.trainable_parameters.hidden, self._graph_data)
        data, hidden = self._graph_name_data, self._graph_name_hidden
        input_dim = self.multi_step * self.num_input_channels + self.node_attributes.attr_ndims[data]

        self.encoder = instantiate(
            model_config.model.encoder,
            in_channels_src=input_dim,
            in_channels_dst=self.node_attributes.attr_ndims[hidden],
            hidden_dim=self.num_channels,
            sub_graph=self._graph_data[(data, "to", hidden)],
            src_grid_size=self.node_attributes.num_nodes[data],
            dst_grid_size=self.node_attributes.num_nodes[hidden],
        )
        self.processor = instantiate(
            model_config.model.processor,
            num_channels=self.num_channels,
            sub_graph=self._graph_data[(hidden, "to", hidden)],
            src_grid_size=self.node_attributes.num_nodes[hidden],
            dst_grid_size=self.node_attributes.num_nodes[hidden],
        )
        self.decoder = instantiate(
            model_config.model.decoder,
            in_channels_src=self.num_channels,
            in_channels_dst=input_dim,
            hidden_dim=self.num_channels,
            out_channels_dst=self.num_output_channels,
            sub_graph=self._graph_data[(hidden, "to", data)],
            src_grid_size=self.node_attributes.num_nodes[hidden],
            dst_grid_size=self.node_attributes.num_nodes[data],
        )
        self.boundings = nn.ModuleList(
            [
                instantiate(cfg, name_to_index=self.data_indices.internal_model.output.name_to_index)
                for cfg in getattr(model_config.model, "bounding", [])
            ]
        )
        self._idx_cache: dict = {}

    def _calculate_shapes_and_indices(self, data_indices) -> None:
        self.num_input_channels = len(data_indices.internal_model.input)
        self.num_output_channels = len(data_indices.internal_model.output)
        self._internal_input_idx = data_indices.internal_model.input.prognostic
        self._internal_output_idx = data_indices.internal_model.output.prognostic

    def _assert_matching_indices(self, data_indices) -> None:
        n_full = len(data_indices.internal_model.output.full)
        n_diag = len(data_indices.internal_model.output.diagnostic)
        assert len(self._internal_output_idx) == n_full - n_diag, (
            f"Mismatch between the internal data indices ({len(self._internal_output_idx)}) and "
            f"the internal output indices excluding diagnostic variables ({n_full - n_diag})"
        )
        assert len(self._internal_input_idx) == len(
            self._internal_output_idx
        ), f"Internal model indices must match {self._internal_input_idx} != {self._internal_output_idx}"

    def _run_mapper(self, mapper: nn.Module, data, batch_size: int, shard_shapes, model_comm_group=None,
                    use_reentrant: bool = False):
        """One mapper call as the reference makes it (models/encoder_processor_decoder.py:127-165): under activation
        checkpointing when an autograd graph is being built.  ``forward`` does not come through here -- it drives the
        mappers' ``native`` entry points on padded / re-ordered rows; this is for callers that compose the sub-modules
        themselves (reference-style subclasses)."""
        kwargs = dict(batch_size=batch_size, shard_shapes=shard_shapes, model_comm_group=model_comm_group)
        if not torch.is_grad_enabled():
            return mapper(data, **kwargs)
        from torch.utils.checkpoint import checkpoint

        return checkpoint(mapper, data, **kwargs, use_reentrant=use_reentrant)

    def _prognostic_indices(self, device):
        key = str(device)
        if key not in self._idx_cache:
            as_i32 = lambda t: torch.as_tensor(t).to(device=device, dtype=torch.int32).contiguous()  # noqa: E731
            self._idx_cache[key] = (as_i32(self._internal_output_idx), as_i32(self._internal_input_idx))
        return self._idx_cache[key]

    def _mesh_order(self, device):
        """(order, inverse) of the internal, locality-preserving mesh row order (cached per device)."""
        key = ("mesh_order", str(device))
        if key not in self._idx_cache:
            latlons = self.node_attributes.latlons(self._graph_name_hidden)
            if not getattr(self, "mesh_locality_order", True):  # tests: the graph's own node order (an invariance check)
                order = torch.arange(latlons.shape[0], device=device)
            else:
                order = runtime.locality_order(latlons).to(device)
            self._idx_cache[key] = (order, runtime.inverse_permutation(order))
        return self._idx_cache[key]

    def _with_ones(self, trainable: Optional[Tensor], n_nodes: int, enabled: bool, device=None) -> Optional[Tensor]:
        """The per-node trainable columns with a constant-1 column appended (``assemble_nodes`` writes them behind the
        coordinates): the bias carrier of the mappers' embedding fold."""
        if not enabled:
            return trainable
        device = trainable.device if trainable is not None else (device or self.node_attributes.latlons(
            self._graph_name_data).device)
        key = ("ones", n_nodes, str(device))
        if key not in self._idx_cache:
            self._idx_cache[key] = torch.ones((n_nodes, 1), dtype=torch.float32, device=device)
        ones = self._idx_cache[key]
        return ones if trainable is None else torch.cat([trainable.detach().float(), ones], dim=1)

    def _embed_fold(self, dtype: torch.dtype) -> bool:
        """Both mappers are GraphTransformer mappers that take feature matrices with a constant-1 padding column."""
        from ..layers.mapper import GraphTransformerBaseMapper

        return (runtime.embed_fold_enabled(dtype) and isinstance(self.encoder, GraphTransformerBaseMapper)
                and isinstance(self.decoder, GraphTransformerBaseMapper))

    @staticmethod
    def _feature_ld(width: int, dtype: torch.dtype, wide: bool = True) -> int:
        """Row pitch of an assembled feature matrix: the K-slab multiple, and (``wide``, bf16) at least two slabs: the
        persistent MFMA kernel takes K >= 128, one slab of zero columns is cheaper than the generic kernel."""
        ld = ops.round_up(width, ops.k_multiple(dtype))
        return max(ld, 128) if (wide and dtype == torch.bfloat16) else ld

    @staticmethod
    def _one_cols(mapper, src: Optional[int], dst: Optional[int]) -> dict:
        """``one_cols`` keyword of the GraphTransformer mappers' ``native`` (other mapper families do not take it)."""
        from ..layers.mapper import GraphTransformerBaseMapper

        return {"one_cols": (src, dst)} if isinstance(mapper, GraphTransformerBaseMapper) else {}

    def _finish(self, y: Tensor, x: Tensor, input_affine=None, output_affine=None, rows: Optional[Tensor] = None) -> Tensor:
        """Prognostic residual, boundings, optional de-normalisation (reference :223-233 + the interface's
        post-processor when it is a plain InputNormalizer).  ``rows`` (int64 grid ids, batch 1 / ensemble 1): ``y`` holds
        only those nodes' rows ``[1, 1, len(rows), V_out]`` -- every step here is row-local, so a rank of a node-partitioned
        run finishes the rows it decoded before they are all-gathered."""
        key = ("residual_src", str(y.device))
        if key not in self._idx_cache:
            src = torch.full((self.num_output_channels,), -1, dtype=torch.int32)
            src[torch.as_tensor(self._internal_output_idx).long()] = torch.as_tensor(self._internal_input_idx).to(torch.int32)
            self._idx_cache[key] = src.to(y.device)
        if len(self.boundings) == 0:
            ops.finalize_output(y, x, self._idx_cache[key], input_affine, output_affine, rows=rows)
            return y if y.dtype == x.dtype else y.to(x.dtype)
        plan = self._bounding_plan(y.device, output_affine)
        if plan is None:  # a bounding class this package does not know: call the modules, as the reference does
            ops.finalize_output(y, x, self._idx_cache[key], input_affine, None, rows=rows)
            if y.dtype != x.dtype:
                y = y.to(x.dtype)
            for bounding in self.boundings:
                y = bounding(y)
            if output_affine is not None:  # boundings act on the normalised output: de-normalise after them
                y = (y - output_affine[1]) / output_affine[0]
            return y
        # known boundings: the columns they touch stay normalised through finalize_output (mul 1 / add 0 there) and
        # are bounded, then de-normalised, by ONE kernel; every other column is finished by finalize_output
        op_lists, masked_affine, fin = plan
        ops.finalize_output(y, x, self._idx_cache[key], input_affine, masked_affine, rows=rows)
        ops.bound_output(y, *op_lists, fin=fin)
        return y if y.dtype == x.dtype else y.to(x.dtype)

    def _bounding_plan(self, device, output_affine):
        """Device-side op list of ``self.boundings`` (cached per device and output affine), ``None`` for unknown classes."""
        from ..layers.bounding import bounded_columns
        from ..layers.bounding import compile_boundings

        key = ("boundings", str(device), None if output_affine is None else
               (output_affine[0].data_ptr(), output_affine[0]._version, output_affine[1].data_ptr(),
                output_affine[1]._version))
        if key not in self._idx_cache:
            op_list = compile_boundings(self.boundings)
            if op_list is None:
                self._idx_cache[key] = None
                return None
            lists = (torch.tensor([o[0] for o in op_list], dtype=torch.int32, device=device),
                     torch.tensor([o[1] for o in op_list], dtype=torch.float32, device=device),
                     torch.tensor([o[2] for o in op_list], dtype=torch.float32, device=device),
                     torch.tensor([o[3] for o in op_list], dtype=torch.int32, device=device))
            masked = fin = None
            if output_affine is not None:
                cols = torch.tensor(bounded_columns(op_list), dtype=torch.int64, device=device)
                mul, add = (t.detach().to(device=device, dtype=torch.float32).contiguous() for t in output_affine)
                m_mul, m_add = mul.clone(), add.clone()
                m_mul[cols] = 1.0
                m_add[cols] = 0.0
                masked = (m_mul, m_add)
                fin = (cols.to(torch.int32), mul[cols].contiguous(), add[cols].contiguous())
            self._idx_cache[key] = (lists, masked, fin)
        return self._idx_cache[key]

    def _training_forward(self, x: Tensor, input_affine=None, output_affine=None) -> Tensor:
        """Forward WITH an autograd graph (anemoi-training calls ``.backward()`` on a loss of the result): the reference's
        own composition -- encoder, processor, skip, decoder, residual, boundings -- through the sub-modules' ``forward``
        methods, which take their differentiable routes (``training.py``: every heavy op and its backward on the HIP
        kernels, mappers and processor chunks checkpointed as in the reference).  SURVEY section 8f-1."""
        from .. import training

        if input_affine is not None or output_affine is not None:
            raise NotImplementedError("input_affine / output_affine belong to the inference interface (predict_step)")
        return training.model_forward(self, x)

    def forward(self, x: Tensor, model_comm_group=None, *, input_affine=None, output_affine=None) -> Tensor:
        """``input_affine`` / ``output_affine`` (keyword-only extension, ``(mul, add)`` per input / output variable): ``x``
        is the RAW state and the result is de-normalised -- ``InputNormalizer`` folded into the first and the last
        kernel of the forward (``AnemoiModelInterface.predict_step`` uses it); default: the reference's contract."""
        if model_comm_group is not None and model_comm_group.size() > 1:
            from ..distributed.partition import sharded_forward

            return sharded_forward(self, x, model_comm_group, input_affine=input_affine, output_affine=output_affine)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return self._training_forward(x, input_affine, output_affine)
        runtime.require_inference(self)
        batch_size, _, ensemble_size, grid, _ = x.shape
        dtype = runtime.compute_dtype(x)
        kmult = ops.k_multiple(dtype)
        data, hidden = self._graph_name_data, self._graph_name_hidden
        na = self.node_attributes

        # [x (time-major) | sin/cos latlon | trainable | 1 | 0-pad]: written once, straight into the GEMM input layout.
        # The constant 1 in the first padding column carries the embedding bias when a GraphTransformer mapper folds
        # its embedding into the block's first GEMMs (layers/mapper.py::_embedded); zero weights meet it otherwise.
        width = self.multi_step * self.num_input_channels + na.attr_ndims[data]
        fold = self._embed_fold(dtype)
        tr_data = na.trainable_tensors[data].trainable
        x_data = ops.assemble_nodes(x, na.latlons(data), self._with_ones(tr_data, grid, fold), batch_size, dtype,
                                    ld_out=self._feature_ld(width + int(fold), dtype, fold), in_affine=input_affine)
        # mesh rows live in an internal Morton order (gather locality of the edge kernels); only the tiny
        # per-node attribute tables are permuted, the mesh never leaves the model
        order, inv = self._mesh_order(x.device)
        tr_hidden = na.trainable_tensors[hidden].trainable
        w_hidden = na.attr_ndims[hidden]
        x_hidden = ops.assemble_nodes(None, na.latlons(hidden)[order],
                                      self._with_ones(None if tr_hidden is None else tr_hidden[order], order.numel(), fold),
                                      batch_size, dtype, ld_out=self._feature_ld(w_hidden + int(fold), dtype, fold))
        one_data, one_hidden = (width, w_hidden) if fold else (None, None)

        enc = self.encoder.native(x_data, x_hidden, batch_size, dst_map=inv,
                                  **self._one_cols(self.encoder, one_data, one_hidden))
        # GraphTransformer forward mapper hands the RAW data input on to the decoder (reference layers/mapper.py:345);
        # the GNN forward mapper hands on its UPDATED source embedding (reference layers/mapper.py:522)
        x_data_latent, x_latent = (x_data, enc) if not isinstance(enc, tuple) else enc
        x_proc = self.processor.native(x_latent, batch_size, node_map=inv)
        x_latent_proc = ops.add(x_proc, x_latent)
        y = self.decoder.native(x_latent_proc, x_data_latent, batch_size, out_dtype=torch.float32, src_map=inv,
                                **self._one_cols(self.decoder, None, one_data if x_data_latent is x_data else None))
        if isinstance(y, tuple):
            y = y[1]

        y = y.view(batch_size, ensemble_size, grid, self.num_output_channels)
        return self._finish(y, x, input_affine, output_affine)
