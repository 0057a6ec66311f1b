"""Stand-ins for the third-party packages the reference imports but this image lacks.

Used ONLY by ``tests/golden/make_golden.py`` in the build container to import
``/root/reference/src/anemoi/models`` as-is and record golden vectors.  This is
our own code (no reference source); it never runs on the GPU box (the fixtures
it produced are what travels).

Each stand-in restates the documented contract of the real package for exactly
the calls the reference makes:

* ``torch_geometric.nn.conv.MessagePassing`` -- ``propagate`` with a Tensor
  ``edge_index`` and ``flow='source_to_target'``: ``*_j`` arguments are gathered
  with ``edge_index[0]``, ``*_i`` with ``edge_index[1]``; ``index=edge_index[1]``;
  ``ptr=None``; ``size_i = dim_size = size[1]`` (inferred from the ``_i`` tensor
  when ``size`` is None); default aggregation = scatter-sum over dim 0.
* ``torch_geometric.utils.{scatter, softmax, mask_to_index, k_hop_subgraph,
  bipartite_subgraph}`` -- PyG 2.4 semantics, written independently of
  ``oracle/pyg_semantics.py`` (index_add / python loops) so the two restatements
  check each other.
* ``torch_geometric.data.HeteroData`` -- dict-of-stores.
* ``hydra.utils.instantiate`` -- ``_target_`` import + kwargs merge.
* ``anemoi.utils.config.DotDict`` -- attribute-access dict.
"""

from __future__ import annotations

import importlib
import inspect
import sys
import types
from typing import Optional

import torch
from torch import Tensor


# ---------------------------------------------------------------- utils
def scatter(src: Tensor, index: Tensor, dim: int = 0, dim_size: Optional[int] = None, reduce: str = "sum") -> Tensor:
    assert dim == 0
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() else 0
    shape = (dim_size,) + tuple(src.shape[1:])
    if reduce in ("sum", "add"):
        return src.new_zeros(shape).index_add_(0, index, src)
    if reduce in ("max", "amax"):
        out = src.new_zeros(shape)
        idx = index.view((-1,) + (1,) * (src.dim() - 1)).expand_as(src)
        return out.scatter_reduce_(0, idx, src, reduce="amax", include_self=False)
    raise ValueError(reduce)


def softmax(src: Tensor, index: Tensor, ptr=None, num_nodes: Optional[int] = None, dim: int = 0) -> Tensor:
    assert ptr is None and dim == 0
    n = num_nodes if num_nodes is not None else int(index.max()) + 1
    src_max = scatter(src.detach(), index, 0, n, reduce="max")
    out = (src - src_max.index_select(0, index)).exp()
    out_sum = scatter(out, index, 0, n, reduce="sum") + 1e-16
    return out / out_sum.index_select(0, index)


def mask_to_index(mask: Tensor) -> Tensor:
    return mask.nonzero(as_tuple=False).view(-1)


def k_hop_subgraph(node_idx, num_hops, edge_index, relabel_nodes=False, num_nodes=None, flow="source_to_target",
                   directed=False):
    assert num_hops == 1 and directed and not relabel_nodes and flow == "source_to_target"
    if num_nodes is None:
        num_nodes = int(edge_index.max()) + 1
    row, col = edge_index  # source_to_target: (row, col) = (src, dst); targets are `col`
    node_mask = row.new_zeros(num_nodes, dtype=torch.bool)
    node_mask[node_idx] = True
    edge_mask = node_mask[col]
    subset = torch.cat([node_idx, row[edge_mask]]).unique()
    return subset, edge_index[:, edge_mask], None, edge_mask


def bipartite_subgraph(subset, edge_index, edge_attr=None, relabel_nodes=False, size=None, return_edge_mask=False):
    assert not relabel_nodes
    src_subset, dst_subset = subset
    src_mask = torch.zeros(size[0], dtype=torch.bool, device=edge_index.device)
    dst_mask = torch.zeros(size[1], dtype=torch.bool, device=edge_index.device)
    src_mask[src_subset] = True
    dst_mask[dst_subset] = True
    edge_mask = src_mask[edge_index[0]] & dst_mask[edge_index[1]]
    ei = edge_index[:, edge_mask]
    ea = edge_attr[edge_mask] if edge_attr is not None else None
    return ei, ea


# ---------------------------------------------------------------- MessagePassing
class MessagePassing(torch.nn.Module):
    def __init__(self, aggr: str = "add", flow: str = "source_to_target", node_dim: int = -2, **kwargs):
        super().__init__()
        assert flow == "source_to_target"
        self.aggr = aggr
        self.node_dim = node_dim

    def propagate(self, edge_index: Tensor, size=None, **kwargs):
        size = [None, None] if size is None else list(size)
        msg_params = [p for p in inspect.signature(self.message).parameters]
        coll = {}
        for name in msg_params:
            if name.endswith("_i") or name.endswith("_j"):
                if name[:-2] not in kwargs:
                    continue
                side = 1 if name.endswith("_i") else 0
                data = kwargs[name[:-2]]
                if isinstance(data, (tuple, list)):
                    data = data[side]
                if isinstance(data, Tensor):
                    n = data.size(0)
                    if size[side] is None:
                        size[side] = n
                    elif size[side] != n:
                        raise ValueError(f"Encountered tensor with size {n} in dimension 0, but expected {size[side]}")
                    data = data.index_select(0, edge_index[side])
                coll[name] = data
        for k, v in kwargs.items():
            coll.setdefault(k, v)
        coll["index"] = edge_index[1]
        coll["ptr"] = None
        coll["edge_index"] = edge_index
        coll["size_i"] = size[1] if size[1] is not None else size[0]
        coll["dim_size"] = coll["size_i"]
        out = self.message(**{k: coll[k] for k in msg_params})
        if type(self).aggregate is not MessagePassing.aggregate:
            agg_params = list(inspect.signature(self.aggregate).parameters)
            first = agg_params[0]
            return self.aggregate(out, **{k: coll[k] for k in agg_params[1:] if k in coll and k != first})
        return self.aggregate(out, coll["index"], dim_size=coll["dim_size"])

    def aggregate(self, inputs: Tensor, index: Tensor, ptr=None, dim_size=None) -> Tensor:
        assert self.aggr in ("add", "sum")
        return scatter(inputs, index, 0, dim_size, reduce="sum")

    def message(self, x_j):  # pragma: no cover
        return x_j


# ---------------------------------------------------------------- HeteroData
class _Store(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    @property
    def num_nodes(self):
        return self["x"].shape[0]


class HeteroData:
    def __init__(self):
        self._stores = {}

    def __getitem__(self, key):
        if key not in self._stores:
            self._stores[key] = _Store()
        return self._stores[key]

    def __bool__(self):
        return True

    @property
    def node_types(self):
        return [k for k in self._stores if isinstance(k, str)]

    def node_items(self):
        return [(k, v) for k, v in self._stores.items() if isinstance(k, str)]


# ---------------------------------------------------------------- hydra / anemoi.utils
def instantiate(config, *args, **kwargs):
    cfg = dict(config)
    target = cfg.pop("_target_")
    cfg.pop("_convert_", None)
    cfg.pop("_recursive_", None)
    kwargs.pop("_recursive_", None)
    mod, name = target.rsplit(".", 1)
    cls = getattr(importlib.import_module(mod), name)
    cfg.update(kwargs)
    return cls(*args, **cfg)


class DotDict(dict):
    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        for k, v in list(self.items()):
            if isinstance(v, dict) and not isinstance(v, DotDict):
                self[k] = DotDict(v)
            elif isinstance(v, list):
                self[k] = [DotDict(i) if isinstance(i, dict) else i for i in v]

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


def install(reference_src: str = "/root/reference/src") -> None:
    """Register the stand-ins in ``sys.modules`` and put the reference on ``sys.path``."""

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    from typing import Tuple, Union

    pg = mod("torch_geometric")
    pg.typing = mod(
        "torch_geometric.typing",
        Adj=Tensor,
        OptTensor=Optional[Tensor],
        PairTensor=Tuple[Tensor, Tensor],
        OptPairTensor=Tuple[Tensor, Optional[Tensor]],
        Size=Optional[Tuple[int, int]],
    )
    pg.utils = mod(
        "torch_geometric.utils",
        scatter=scatter,
        softmax=softmax,
        mask_to_index=mask_to_index,
        k_hop_subgraph=k_hop_subgraph,
        bipartite_subgraph=bipartite_subgraph,
    )
    pg.nn = mod("torch_geometric.nn")
    pg.nn.conv = mod("torch_geometric.nn.conv", MessagePassing=MessagePassing)
    pg.data = mod("torch_geometric.data", HeteroData=HeteroData)
    hy = mod("hydra")
    hy.utils = mod("hydra.utils", instantiate=instantiate)
    au = mod("anemoi.utils")
    au.__path__ = []
    au.config = mod("anemoi.utils.config", DotDict=DotDict)
    # omegaconf: the reference's IndexCollection only calls OmegaConf.to_container on plain (DotDict) configs
    def _to_container(cfg, resolve=True):
        if isinstance(cfg, dict):
            return {k: _to_container(v) for k, v in cfg.items()}
        if isinstance(cfg, (list, tuple)):
            return [_to_container(v) for v in cfg]
        return cfg

    mod("omegaconf", OmegaConf=type("OmegaConf", (), {"to_container": staticmethod(_to_container)}))
    _ = Union
    if reference_src not in sys.path:
        sys.path.insert(0, reference_src)
