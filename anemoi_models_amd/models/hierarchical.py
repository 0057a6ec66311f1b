"""Hierarchical encoder - processor - decoder mirroring reference models/hierarchical.py:28-308.

Same constructor, same sub-module names (``node_attributes``, ``encoder``, ``down_level_processor``,
``up_level_processor``, ``downscale``, ``upscale``, ``decoder``, ``boundings``) and therefore the same ``state_dict``.
The node sets ``graph.hidden = [h_1 ... h_L]`` carry ``num_channels * 2**i`` features; the forward is
``data -> h_1 -> (process, down) ... h_L (process) ... (up, + skip, process) -> h_1 -> data`` and runs on the same HIP
kernels as the flat model (every mapper / processor is one of ``layers/mapper.py`` / ``layers/processor.py``).
"""

from __future__ import annotations


import torch
from torch import Tensor
from torch import nn

from .. import ops
from .. import runtime
from ..layers.graph import NamedNodesAttributes
from .encoder_processor_decoder import AnemoiModelEncProcDec
from .encoder_processor_decoder import instantiate


class AnemoiModelEncProcDecHierarchical(AnemoiModelEncProcDec):
    """Message passing hierarchical graph network on MI355X kernels."""

    def __init__(self, *, model_config, data_indices, graph_data) -> None:
        nn.Module.__init__(self)
        self._graph_data = graph_data
        self._graph_name_data = model_config.graph.data
        self._graph_hidden_names = list(model_config.graph.hidden)
        self.num_hidden = len(self._graph_hidden_names)
        self.level_process = model_config.model.enable_hierarchical_level_processing
        # feature width per depth (reference :61-63)
        self.hidden_dims = {
            hidden: model_config.model.num_channels * (2**i) for i, hidden in enumerate(self._graph_hidden_names)
        }

        self._calculate_shapes_and_indices(data_indices)
        self._assert_matching_indices(data_indices)
        self.data_indices = data_indices
        self.multi_step = model_config.training.multistep_input

        self.node_attributes = NamedNodesAttributes(model_config.model.trainable_parameters.hidden, self._graph_data)
        na, data, names = self.node_attributes, self._graph_name_data, self._graph_hidden_names
        input_dim = self.multi_step * self.num_input_channels + na.attr_ndims[data]

        self.encoder = instantiate(
            model_config.model.encoder,
            in_channels_src=input_dim,
            in_channels_dst=na.attr_ndims[names[0]],
            hidden_dim=self.hidden_dims[names[0]],
            sub_graph=self._graph_data[(data, "to", names[0])],
            src_grid_size=na.num_nodes[data],
            dst_grid_size=na.num_nodes[names[0]],
        )

        if self.level_process:
            self.down_level_processor = nn.ModuleDict()
            self.up_level_processor = nn.ModuleDict()
            for name in names:
                for holder in (self.down_level_processor, self.up_level_processor):
                    holder[name] = instantiate(
                        model_config.model.processor,
                        num_channels=self.hidden_dims[name],
                        sub_graph=self._graph_data[(name, "to", name)],
                        src_grid_size=na.num_nodes[name],
                        dst_grid_size=na.num_nodes[name],
                        num_layers=model_config.model.level_process_num_layers,
                    )
            # the coarsest level is processed once: |->|->|<-|<-|  (reference :113-114)
            del self.up_level_processor[names[-1]]

        self.downscale = nn.ModuleDict()
        for src, dst in zip(names[:-1], names[1:]):
            self.downscale[src] = instantiate(
                model_config.model.encoder,
                in_channels_src=self.hidden_dims[src],
                in_channels_dst=na.attr_ndims[dst],
                hidden_dim=self.hidden_dims[dst],
                sub_graph=self._graph_data[(src, "to", dst)],
                src_grid_size=na.num_nodes[src],
                dst_grid_size=na.num_nodes[dst],
            )

        self.upscale = nn.ModuleDict()
        for dst, src in zip(names[:-1], names[1:]):
            self.upscale[src] = instantiate(
                model_config.model.decoder,
                in_channels_src=self.hidden_dims[src],
                in_channels_dst=self.hidden_dims[dst],
                hidden_dim=self.hidden_dims[src],
                out_channels_dst=self.hidden_dims[dst],
                sub_graph=self._graph_data[(src, "to", dst)],
                src_grid_size=na.num_nodes[src],
                dst_grid_size=na.num_nodes[dst],
            )

        self.decoder = instantiate(
            model_config.model.decoder,
            in_channels_src=self.hidden_dims[names[0]],
            in_channels_dst=input_dim,
            hidden_dim=self.hidden_dims[names[0]],
            out_channels_dst=self.num_output_channels,
            sub_graph=self._graph_data[(names[0], "to", data)],
            src_grid_size=na.num_nodes[names[0]],
            dst_grid_size=na.num_nodes[data],
        )
        self.boundings = nn.ModuleList(
            [
                instantiate(cfg, name_to_index=self.data_indices.internal_model.output.name_to_index)
                for cfg in getattr(model_config.model, "bounding", [])
            ]
        )
        self._idx_cache: dict = {}

    def forward(self, x: Tensor, model_comm_group=None) -> Tensor:
        if model_comm_group is not None and model_comm_group.size() > 1:
            # The reference's own composition does not run there either: ``downscale`` is a forward mapper, whose
            # pre-processing splits its source by the FULL shard shapes (layers/mapper.py:108-111 -> torch.split in
            # distributed/primitives.py:49) although the level input is already a shard (models/hierarchical.py:236-244),
            # and the gathered output of ``upscale`` (layers/mapper.py:99-102) is added to the sharded skip (:272).
            raise NotImplementedError("the hierarchical model does not run across a model group (neither does the "
                                      "reference's: its level mappers split already-sharded rows)")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from .. import training

            return training.hierarchical_forward(self, x)
        batch_size, _, ensemble_size, grid, _ = x.shape
        dtype = runtime.compute_dtype(x)
        kmult = ops.k_multiple(dtype)
        data, names, na = self._graph_name_data, self._graph_hidden_names, self.node_attributes

        # [x | coordinates | trainable | 1 | 0-pad]: the constant 1 serves the mappers' embedding fold (see the flat model)
        width = self.multi_step * self.num_input_channels + na.attr_ndims[data]
        fold = self._embed_fold(dtype)
        x_data = ops.assemble_nodes(x, na.latlons(data), self._with_ones(na.trainable_tensors[data].trainable, grid, fold),
                                    batch_size, dtype, ld_out=self._feature_ld(width + int(fold), dtype, fold))
        one_data = width if fold else None
        x_hidden = {
            h: ops.assemble_nodes(None, na.latlons(h), na.trainable_tensors[h].trainable, batch_size, dtype,
                                  ld_out=ops.round_up(na.attr_ndims[h], kmult))
            for h in names
        }

        def first(out):  # GraphTransformer mappers return the destination nodes; GNN mappers (src, dst)
            return out[1] if isinstance(out, tuple) else out

        curr = first(self.encoder.native(x_data, x_hidden[names[0]], batch_size,
                                         **self._one_cols(self.encoder, one_data, None)))
        x_skip, x_encoded = {}, {}
        for src, dst in zip(names[:-1], names[1:]):  # ---- down (reference :224-249)
            if self.level_process:
                curr = self.down_level_processor[src].native(curr, batch_size)
            x_skip[src] = curr
            out = self.downscale[src].native(curr, x_hidden[dst], batch_size)
            # the source the mapper hands back becomes the upscale mapper's destination input: the GraphTransformer
            # mapper returns its raw source (layers/mapper.py:345), the GNN mapper its updated source embedding (:522)
            x_encoded[src], curr = out if isinstance(out, tuple) else (curr, out)
        if self.level_process:  # coarsest level (reference :252-258)
            curr = self.down_level_processor[names[-1]].native(curr, batch_size)
        for dst, src in zip(reversed(names[:-1]), reversed(names[1:])):  # ---- up (reference :261-286)
            curr = first(self.upscale[src].native(curr, x_encoded[dst], batch_size, out_dtype=dtype))
            curr = ops.add(curr, x_skip[dst])
            if self.level_process:
                curr = self.up_level_processor[dst].native(curr, batch_size)

        y = first(self.decoder.native(curr, x_data, batch_size, out_dtype=torch.float32,
                                      **self._one_cols(self.decoder, None, one_data)))
        y = y.view(batch_size, ensemble_size, grid, self.num_output_channels)
        return self._finish(y, x)  # prognostic residual + boundings (reference models/hierarchical.py:300-308)
