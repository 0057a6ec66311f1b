"""Graph ingestion: the ``torch.save``d ``HeteroData`` files anemoi-graphs writes.

anemoi-training loads its graph with ``torch.load(path, weights_only=False)`` and hands the ``HeteroData`` to the model
(reference models/encoder_processor_decoder.py:54-98 only ever reads ``graph[name].x / .num_nodes``,
``graph[(src, "to", dst)].edge_index`` and named edge attributes).  :func:`load_graph` does the same and returns

* the ``HeteroData`` itself when torch-geometric is importable (every module here consumes it duck-typed), or
* this package's :class:`GraphData` rebuilt from the pickle when torch-geometric is NOT installed: the unpickler maps the
  ``torch_geometric.data`` classes to inert records and copies their stores.  The pickle layout followed is PyG 2.3 / 2.4's
  (``HeteroData.__dict__``: ``_global_store``, ``_node_store_dict``, ``_edge_store_dict``; storages: ``_mapping``) --
  restated from the published PyG source, NOT checked against a PyG-written file in this image (PyG is absent here).

:func:`save_graph` writes the PyG-free form (a plain dictionary of tensors) that :func:`load_graph` also reads.
"""

from __future__ import annotations

import pickle
from typing import Any

import torch

from .data import GraphData

_FORMAT = "anemoi_models_amd.GraphData/1"


class _Record:
    """Inert stand-in for a ``torch_geometric`` class met while unpickling: keeps whatever state the pickle carries."""

    def __init__(self, *args, **kwargs) -> None:
        self.__dict__["_args"] = args
        self.__dict__.update(kwargs)

    def __setstate__(self, state) -> None:
        if isinstance(state, dict):
            self.__dict__.update(state)
        else:
            self.__dict__["_state"] = state


class _CompatUnpickler(pickle.Unpickler):
    def find_class(self, module: str, name: str) -> Any:
        if module.split(".")[0] == "torch_geometric":
            return type(name, (_Record,), {"_pyg_module": module})
        return super().find_class(module, name)


class _CompatPickle:
    """``pickle_module`` for ``torch.load``: the stock pickle with ``torch_geometric.*`` classes mapped to records."""

    __name__ = "pickle"
    Unpickler = _CompatUnpickler
    load = staticmethod(pickle.load)
    loads = staticmethod(pickle.loads)
    dump = staticmethod(pickle.dump)
    dumps = staticmethod(pickle.dumps)
    Pickler = pickle.Pickler
    PickleError = pickle.PickleError
    UnpicklingError = pickle.UnpicklingError


def _mapping(store) -> dict:
    d = getattr(store, "__dict__", {})
    m = d.get("_mapping")
    return dict(m) if isinstance(m, dict) else {k: v for k, v in d.items() if not k.startswith("_")}


def _from_records(obj) -> GraphData:
    d = obj.__dict__
    if "_node_store_dict" not in d or "_edge_store_dict" not in d:
        raise ValueError("load_graph: the pickle holds a torch_geometric object that is not a HeteroData")
    g = GraphData()
    for name, store in d["_node_store_dict"].items():
        for k, v in _mapping(store).items():
            g[name][k] = v
    for key, store in d["_edge_store_dict"].items():
        for k, v in _mapping(store).items():
            g[tuple(key)][k] = v
    return g


def _from_plain(d: dict) -> GraphData:
    g = GraphData()
    for name, store in d["nodes"].items():
        for k, v in store.items():
            g[name][k] = v
    for key, store in d["edges"].items():
        for k, v in store.items():
            g[tuple(key.split("|"))][k] = v
    return g


def save_graph(graph, path: str) -> None:
    """Write ``graph`` (``GraphData`` or ``HeteroData``) as a plain dictionary of tensors (no torch-geometric needed to
    read it back)."""
    nodes = {name: {k: v for k, v in store.items() if isinstance(v, torch.Tensor)} for name, store in graph.node_items()}
    edges = {"|".join(key): {k: v for k, v in graph[key].items() if isinstance(v, torch.Tensor)}
             for key in graph.edge_types}
    torch.save({"format": _FORMAT, "nodes": nodes, "edges": edges}, path)


def load_graph(path: str, map_location="cpu"):
    """Load a graph file: an anemoi-graphs ``HeteroData`` pickle or a :func:`save_graph` file."""
    try:
        import torch_geometric  # noqa: F401

        have_pyg = True
    except ImportError:
        have_pyg = False
    if have_pyg:
        obj = torch.load(path, map_location=map_location, weights_only=False)
    else:
        obj = torch.load(path, map_location=map_location, weights_only=False, pickle_module=_CompatPickle)
    if isinstance(obj, dict) and obj.get("format") == _FORMAT:
        return _from_plain(obj)
    if isinstance(obj, _Record):
        return _from_records(obj)
    if hasattr(obj, "node_items") and hasattr(obj, "edge_types"):
        return obj  # a real HeteroData (or a pickled GraphData)
    raise ValueError(f"load_graph: {path} does not hold a graph (got {type(obj).__name__})")
