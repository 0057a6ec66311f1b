"""Minimal graph container, duck-type compatible with ``torch_geometric.data.HeteroData``.

The reference only needs this much of ``HeteroData`` on the forward path
(reference models/encoder_processor_decoder.py:54-98, layers/graph.py:78-88,
layers/mapper.py:141-145):

* ``graph[name].x`` / ``graph[name].num_nodes`` for node sets,
* ``graph[(src, "to", dst)].edge_index`` and ``graph[(src, "to", dst)][attr]`` for edge sets,
* ``graph.node_types`` and ``graph.node_items()``.

A real ``HeteroData`` (e.g. one written by anemoi-graphs) can be passed to every
module of this package instead; nothing here is required by the kernels.
"""

from __future__ import annotations

from typing import Any


class Store(dict):
    """Attribute + item access store (one node set or one edge set)."""

    def __getattr__(self, key: str) -> Any:
        try:
            return self[key]
        except KeyError as e:
            raise AttributeError(key) from e

    def __setattr__(self, key: str, value: Any) -> None:
        self[key] = value

    @property
    def num_nodes(self) -> int:
        return int(self["x"].shape[0])

    def to(self, *args, **kwargs) -> "Store":
        out = Store()
        for k, v in self.items():
            out[k] = v.to(*args, **kwargs) if hasattr(v, "to") else v
        return out


class GraphData:
    """Dictionary of node stores (``str`` keys) and edge stores (``(src, rel, dst)`` keys)."""

    def __init__(self) -> None:
        self._stores: dict = {}

    def __getitem__(self, key) -> Store:
        if key not in self._stores:
            self._stores[key] = Store()
        return self._stores[key]

    def __contains__(self, key) -> bool:
        return key in self._stores

    def __bool__(self) -> bool:
        return True

    @property
    def node_types(self) -> list:
        return [k for k in self._stores if isinstance(k, str)]

    @property
    def edge_types(self) -> list:
        return [k for k in self._stores if isinstance(k, tuple)]

    def node_items(self) -> list:
        return [(k, v) for k, v in self._stores.items() if isinstance(k, str)]

    def edge_items(self) -> list:
        return [(k, v) for k, v in self._stores.items() if isinstance(k, tuple)]

    def to(self, *args, **kwargs) -> "GraphData":
        out = GraphData()
        for k, v in self._stores.items():
            out._stores[k] = v.to(*args, **kwargs)
        return out
