"""Model configurations of BASELINE.json as config dictionaries (the schema anemoi-training feeds to Hydra).

``_target_`` strings use the reference's module paths on purpose: :func:`anemoi_models_amd.utils.config.instantiate`
redirects ``anemoi.models.*`` to this package, which is exactly what a user switching frameworks relies on.
"""

from __future__ import annotations

from .config import DotDict

EDGE_ATTRS = ["edge_length", "edge_dirs"]

# name: (graph, channels, processor layers, heads)
BASELINE_CONFIGS = {
    "cfg1": ("o32_ico2", 64, 4, 16),
    "cfg2": ("o96_ico5", 512, 16, 16),
    "cfg3": ("n320_ico6", 1024, 16, 16),
}


def model_config(processor: str = "GraphTransformer", channels: int = 64, layers: int = 4, heads: int = 16,
                 multistep: int = 2, trainable: int = 8, proc_chunks: int = 2, window_size: int = 512,
                 mappers: str = "GraphTransformer") -> DotDict:
    common = {"sub_graph_edge_attributes": EDGE_ATTRS, "trainable_size": trainable}
    mapper = {"activation": "GELU", "num_chunks": 1, "mlp_hidden_ratio": 4, "num_heads": heads, **common}
    procs = {
        "GraphTransformer": {
            "_target_": "anemoi.models.layers.processor.GraphTransformerProcessor", "activation": "GELU",
            "num_layers": layers, "num_chunks": proc_chunks, "mlp_hidden_ratio": 4, "num_heads": heads, **common,
        },
        "GNN": {
            "_target_": "anemoi.models.layers.processor.GNNProcessor", "activation": "SiLU", "num_layers": layers,
            "num_chunks": proc_chunks, "mlp_extra_layers": 0, **common,
        },
        "Transformer": {
            "_target_": "anemoi.models.layers.processor.TransformerProcessor", "activation": "GELU",
            "num_layers": layers, "num_chunks": proc_chunks, "mlp_hidden_ratio": 4, "num_heads": heads,
            "window_size": window_size, "dropout_p": 0.0,
        },
    }
    enc = {"_target_": "anemoi.models.layers.mapper.GraphTransformerForwardMapper", **mapper}
    dec = {"_target_": "anemoi.models.layers.mapper.GraphTransformerBackwardMapper", **mapper}
    if mappers == "GNN":
        gm = {"activation": "SiLU", "num_chunks": 1, "mlp_extra_layers": 0, **common}
        enc = {"_target_": "anemoi.models.layers.mapper.GNNForwardMapper", **gm}
        dec = {"_target_": "anemoi.models.layers.mapper.GNNBackwardMapper", **gm}
    return DotDict(
        {
            "graph": {"data": "data", "hidden": "hidden"},
            "training": {"multistep_input": multistep},
            "model": {
                "num_channels": channels,
                "trainable_parameters": {"data": trainable, "hidden": trainable},
                "encoder": enc,
                "processor": procs[processor],
                "decoder": dec,
            },
        }
    )


def hierarchical_model_config(channels: int = 64, heads: int = 16, hidden=("hidden_1", "hidden_2"),
                              level_layers: int = 2, level_process: bool = True, multistep: int = 2,
                              trainable: int = 8) -> DotDict:
    """Config of ``AnemoiModelEncProcDecHierarchical`` (reference models/hierarchical.py:40-66 reads ``graph.hidden`` as
    a LIST of node sets, ``model.enable_hierarchical_level_processing`` and ``model.level_process_num_layers``)."""
    cfg = model_config("GraphTransformer", channels, level_layers, heads, multistep, trainable, proc_chunks=1)
    cfg["graph"]["hidden"] = list(hidden)
    cfg["model"]["enable_hierarchical_level_processing"] = level_process
    cfg["model"]["level_process_num_layers"] = level_layers
    return cfg
