"""Plain-PyTorch CPU restatement of the reference encoder-processor-decoder forward.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Every function follows one reference function (cited as ``file:line`` relative
to ``/root/reference/src/anemoi/models``) and operates on a reference
``state_dict`` (same key names) plus plain tensors, so "identical random
weights" means literally the same dictionary.  No torch_geometric, hydra or
anemoi.utils imports: their arithmetic is restated in :mod:`oracle.pyg_semantics`.
"""

from __future__ import annotations

import math
from typing import Mapping, Optional, Sequence

import torch
import torch.nn.functional as F
from torch import Tensor

from .pyg_semantics import bipartite_dst_mask
from .pyg_semantics import k_hop_edge_mask_directed
from .pyg_semantics import scatter_sum
from .pyg_semantics import segment_softmax

SD = Mapping[str, Tensor]

_ACT = {
    "GELU": lambda t: F.gelu(t),  # nn.GELU() default: exact erf form
    "SiLU": F.silu,
    "ReLU": F.relu,
    "Tanh": torch.tanh,
    "Sigmoid": torch.sigmoid,
    "Identity": lambda t: t,
}


def _act(name: str):
    if name not in _ACT:
        raise RuntimeError(f"activation {name} not supported by the oracle")
    return _ACT[name]


def _lin(sd: SD, p: str, x: Tensor) -> Tensor:
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _ln(sd: SD, p: str, x: Tensor) -> Tensor:
    w = sd[p + ".weight"]
    return F.layer_norm(x, (w.shape[0],), w, sd[p + ".bias"], 1e-5)


# --------------------------------------------------------------------------
# integer / index helpers (bit-exact)
# --------------------------------------------------------------------------


def expand_edges(edge_index: Tensor, edge_inc: Tensor, batch_size: int) -> Tensor:
    """layers/mapper.py:150-171 ``GraphEdgeMixin._expand_edges``."""
    return torch.cat([edge_index + i * edge_inc for i in range(batch_size)], dim=1)


def trainable_tensor(x: Tensor, trainable: Optional[Tensor], batch_size: int) -> Tensor:
    """layers/graph.py:37-44 ``TrainableTensor.forward`` (einops.repeat == Tensor.repeat on dim 0)."""
    latent = [x.repeat(batch_size, 1)]
    if trainable is not None:
        latent.append(trainable.repeat(batch_size, 1))
    return torch.cat(latent, dim=-1)


def get_shape_shards(t: Tensor, dim: int, comm_size: int = 1) -> list:
    """distributed/shapes.py:19-24."""
    return [list(x.shape) for x in torch.tensor_split(t, comm_size, dim=dim)]


def sort_edges_1hop_chunks(num_nodes, edge_attr: Tensor, edge_index: Tensor, num_chunks: int):
    """distributed/khop_edges.py:88-130: stable partition of edges by contiguous dst ranges."""
    n_dst = num_nodes if isinstance(num_nodes, int) else num_nodes[1]
    node_chunks = torch.arange(n_dst, device=edge_index.device).tensor_split(num_chunks)
    edge_attr_list, edge_index_list = [], []
    for chunk in node_chunks:
        if isinstance(num_nodes, int):
            mask = k_hop_edge_mask_directed(chunk, edge_index, n_dst)
        else:
            mask = bipartite_dst_mask(chunk, edge_index, n_dst)
        edge_index_list.append(edge_index[:, mask])
        edge_attr_list.append(edge_attr[mask])
    return edge_attr_list, edge_index_list


# --------------------------------------------------------------------------
# GraphTransformer conv / blocks
# --------------------------------------------------------------------------


def gt_conv(query: Tensor, key: Tensor, value: Tensor, edges: Tensor, edge_index: Tensor, n_dst: int,
            dropout_p: float = 0.0, keep: Optional[Tensor] = None) -> Tensor:
    """layers/conv.py:98-142 ``GraphTransformerConv`` (+ PyG propagate, flow source_to_target).

    query [N_dst,H,D], key/value [N_src,H,D], edges [E,H,D], edge_index int64 [2,E]
    (row 0 = src ``j``, row 1 = dst ``i``).  ``dropout_p`` with ``keep [E, H]`` of 0 / 1: conv.py:140 in training mode,
    ``torch.nn.functional.dropout(alpha, p)`` = ``alpha * keep / (1 - p)`` with the Bernoulli draw handed in (torch's own
    draw cannot be reproduced outside torch; the HIP kernels draw theirs from a counter hash, restated in the tests).
    """
    src, dst = edge_index[0], edge_index[1]
    d = query.shape[-1]
    query_i = query.index_select(0, dst)
    key_j = key.index_select(0, src) + edges  # conv.py:134-135
    value_j = value.index_select(0, src)
    alpha = (query_i * key_j).sum(dim=-1) / d**0.5  # conv.py:137
    alpha = segment_softmax(alpha, dst, n_dst)  # conv.py:139
    if dropout_p > 0.0:  # conv.py:140
        alpha = alpha * keep.to(alpha.dtype) * (1.0 / (1.0 - dropout_p) if dropout_p < 1.0 else 0.0)
    msg = (value_j + edges) * alpha.unsqueeze(-1)  # conv.py:142
    return scatter_sum(msg, dst, n_dst)  # aggr="add", conv.py:92


def _gt_tail(sd: SD, p: str, out: Tensor, x_r: Tensor, x_skip: Tensor, act: str, num_chunks: int = 1) -> Tensor:
    """projection(out + x_r) + skip; node_dst_mlp(.) + .  (block.py:530-538 / :630-633)."""
    a = _act(act)
    out = torch.cat([_lin(sd, p + ".projection", c) for c in torch.tensor_split(out + x_r, num_chunks, dim=0)], dim=0)
    out = out + x_skip
    res = []
    for c in out.tensor_split(num_chunks, dim=0):
        h = _ln(sd, p + ".node_dst_mlp.0", c)
        h = a(_lin(sd, p + ".node_dst_mlp.1", h))
        res.append(_lin(sd, p + ".node_dst_mlp.3", h) + c)
    return torch.cat(res, dim=0)


def gt_processor_block(
    sd: SD, p: str, x: Tensor, edge_attr: Tensor, edge_index: Tensor, num_heads: int, act: str = "GELU"
) -> Tensor:
    """layers/block.py:602-635 ``GraphTransformerProcessorBlock.forward`` (no comm group)."""
    n, c = x.shape
    d = sd[p + ".lin_query.weight"].shape[0] // num_heads
    x_skip = x
    xh = _ln(sd, p + ".layer_norm1", x)
    x_r = _lin(sd, p + ".lin_self", xh)
    q = _lin(sd, p + ".lin_query", xh).view(n, num_heads, d)
    k = _lin(sd, p + ".lin_key", xh).view(n, num_heads, d)
    v = _lin(sd, p + ".lin_value", xh).view(n, num_heads, d)
    e = _lin(sd, p + ".lin_edge", edge_attr).view(-1, num_heads, d)
    out = gt_conv(q, k, v, e, edge_index, n).reshape(n, num_heads * d)
    return _gt_tail(sd, p, out, x_r, x_skip, act)


def gt_mapper_block(
    sd: SD,
    p: str,
    x_src: Tensor,
    x_dst: Tensor,
    edge_attr: Tensor,
    edge_index: Tensor,
    num_heads: int,
    act: str = "GELU",
    num_chunks: int = 1,
) -> Tensor:
    """layers/block.py:479-550 ``GraphTransformerMapperBlock.forward`` (update_src_nodes=False).

    ``num_chunks`` > 1 follows the inference chunking branch (block.py:508-524):
    edges partitioned by contiguous dst ranges, conv per chunk, results summed.
    Returns the new dst nodes (src nodes are returned unchanged by the reference).
    """
    n_src, n_dst = x_src.shape[0], x_dst.shape[0]
    d = sd[p + ".lin_query.weight"].shape[0] // num_heads
    xs = _ln(sd, p + ".layer_norm1", x_src)
    xd = _ln(sd, p + ".layer_norm2", x_dst)
    x_r = _lin(sd, p + ".lin_self", xd)
    q = _lin(sd, p + ".lin_query", xd).view(n_dst, num_heads, d)
    k = _lin(sd, p + ".lin_key", xs).view(n_src, num_heads, d)
    v = _lin(sd, p + ".lin_value", xs).view(n_src, num_heads, d)
    e = _lin(sd, p + ".lin_edge", edge_attr).view(-1, num_heads, d)
    if num_chunks > 1:
        ea_list, ei_list = sort_edges_1hop_chunks((n_src, n_dst), e, edge_index, num_chunks)
        out = torch.zeros((n_dst, num_heads, d))
        for ea, ei in zip(ea_list, ei_list):
            out += gt_conv(q, k, v, ea, ei, n_dst)
    else:
        out = gt_conv(q, k, v, e, edge_index, n_dst)
    out = out.reshape(n_dst, num_heads * d)
    return _gt_tail(sd, p, out, x_r, x_dst, act, num_chunks)


def gt_forward_mapper(
    sd: SD, p: str, x_src: Tensor, x_dst: Tensor, edge_attr_buf: Tensor, edge_index_base: Tensor,
    batch_size: int, num_heads: int, act: str = "GELU", num_chunks: int = 1,
):
    """layers/mapper.py:275-345 + :245-272 + :108-116 ``GraphTransformerForwardMapper.forward``."""
    edge_attr = trainable_tensor(edge_attr_buf, sd.get(p + ".trainable.trainable"), batch_size)
    edge_index = expand_edges(edge_index_base, sd[p + ".edge_inc"], batch_size)
    xs = _lin(sd, p + ".emb_nodes_src", x_src)
    xd = _lin(sd, p + ".emb_nodes_dst", x_dst)
    out = gt_mapper_block(sd, p + ".proc", xs, xd, edge_attr, edge_index, num_heads, act, num_chunks)
    return x_src, out  # mapper.py:344-345: raw src tensor is handed back


def gt_backward_mapper(
    sd: SD, p: str, x_src: Tensor, x_dst: Tensor, edge_attr_buf: Tensor, edge_index_base: Tensor,
    batch_size: int, num_heads: int, act: str = "GELU", num_chunks: int = 1,
) -> Tensor:
    """layers/mapper.py:348-418 + :96-102 ``GraphTransformerBackwardMapper.forward``."""
    edge_attr = trainable_tensor(edge_attr_buf, sd.get(p + ".trainable.trainable"), batch_size)
    edge_index = expand_edges(edge_index_base, sd[p + ".edge_inc"], batch_size)
    xd = _lin(sd, p + ".emb_nodes_dst", x_dst)  # mapper.py:412-418: only dst is embedded
    out = gt_mapper_block(sd, p + ".proc", x_src, xd, edge_attr, edge_index, num_heads, act, num_chunks)
    out = _ln(sd, p + ".node_data_extractor.0", out)
    return _lin(sd, p + ".node_data_extractor.1", out)


def gt_processor(
    sd: SD, p: str, x: Tensor, edge_attr_buf: Tensor, edge_index_base: Tensor, batch_size: int,
    num_layers: int, num_chunks: int, num_heads: int, act: str = "GELU", return_all: bool = False,
):
    """layers/processor.py:317-343 + layers/chunk.py:225-238 ``GraphTransformerProcessor.forward``."""
    edge_attr = trainable_tensor(edge_attr_buf, sd.get(p + ".trainable.trainable"), batch_size)
    edge_index = expand_edges(edge_index_base, sd[p + ".edge_inc"], batch_size)
    per_chunk = num_layers // num_chunks
    outs = []
    for c in range(num_chunks):
        for b in range(per_chunk):
            x = gt_processor_block(sd, f"{p}.proc.{c}.blocks.{b}", x, edge_attr, edge_index, num_heads, act)
            outs.append(x)
    return (x, outs) if return_all else x


# --------------------------------------------------------------------------
# GNN (edge-MLP message passing) path
# --------------------------------------------------------------------------


def mlp(sd: SD, p: str, x: Tensor, act: str = "SiLU", n_extra_layers: int = 0, layer_norm: bool = True,
        final_activation: bool = False) -> Tensor:
    """layers/mlp.py:74-89 ``MLP``: Linear,act,(Linear,act)x(extra+1),Linear,[act],[LayerNorm]."""
    a = _act(act)
    idx = 0
    x = a(_lin(sd, f"{p}.model.{idx}", x))
    idx += 2
    for _ in range(n_extra_layers + 1):
        x = a(_lin(sd, f"{p}.model.{idx}", x))
        idx += 2
    x = _lin(sd, f"{p}.model.{idx}", x)
    idx += 1
    if final_activation:
        x = a(x)
        idx += 1
    if layer_norm:
        x = _ln(sd, f"{p}.model.{idx}", x).type_as(x)  # layers/utils.py:27-39 AutocastLayerNorm: cast back to the input type
    return x


def gnn_conv(sd: SD, p: str, x_src: Tensor, x_dst: Tensor, edge_attr: Tensor, edge_index: Tensor,
             act: str = "SiLU", n_extra_layers: int = 0):
    """layers/conv.py:61-76 ``GraphConv``: e' = MLP(cat[x_i, x_j, e]) + e ; out = scatter_sum_dst(e')."""
    x_i = x_dst.index_select(0, edge_index[1])
    x_j = x_src.index_select(0, edge_index[0])
    edges_new = mlp(sd, p + ".edge_mlp", torch.cat([x_i, x_j, edge_attr], dim=1), act, n_extra_layers) + edge_attr
    return scatter_sum(edges_new, edge_index[1], x_dst.shape[0]), edges_new


def gnn_processor_block(sd: SD, p: str, x: Tensor, edge_attr: Tensor, edge_index: Tensor, act: str = "SiLU",
                        n_extra_layers: int = 0):
    """layers/block.py:193-223 ``GraphConvProcessorBlock.forward`` (num_chunks=1, no comm group)."""
    out, edges_new = gnn_conv(sd, p + ".conv", x, x, edge_attr, edge_index, act, n_extra_layers)
    nodes_new = mlp(sd, p + ".node_mlp", torch.cat([x, out], dim=1), act, n_extra_layers) + x
    return nodes_new, edges_new


def gnn_processor(sd: SD, p: str, x: Tensor, edge_attr_buf: Tensor, edge_index_base: Tensor, batch_size: int,
                  num_layers: int, num_chunks: int, act: str = "SiLU", n_extra_layers: int = 0) -> Tensor:
    """layers/processor.py:228-250 + layers/chunk.py:165-181 ``GNNProcessor.forward``."""
    edge_attr = trainable_tensor(edge_attr_buf, sd.get(p + ".trainable.trainable"), batch_size)
    edge_index = expand_edges(edge_index_base, sd[p + ".edge_inc"], batch_size)
    per_chunk = num_layers // num_chunks
    for c in range(num_chunks):
        x = x * 1.0
        if c == 0:
            edge_attr = mlp(sd, f"{p}.proc.0.emb_edges", edge_attr, act, n_extra_layers)
        for b in range(per_chunk):
            x, edge_attr = gnn_processor_block(sd, f"{p}.proc.{c}.blocks.{b}", x, edge_attr, edge_index, act,
                                               n_extra_layers)
    return x


def gnn_mapper_block(sd: SD, p: str, x_src: Tensor, x_dst: Tensor, edge_attr: Tensor, edge_index: Tensor,
                     update_src_nodes: bool, act: str = "SiLU", n_extra_layers: int = 0):
    """layers/block.py:249-286 ``GraphConvMapperBlock.forward`` (num_chunks=1, no comm group)."""
    out, edges_new = gnn_conv(sd, p + ".conv", x_src, x_dst, edge_attr, edge_index, act, n_extra_layers)
    new_dst = mlp(sd, p + ".node_mlp", torch.cat([x_dst, out], dim=1), act, n_extra_layers) + x_dst
    new_src = x_src
    if update_src_nodes:  # block.py:282: the source update sees cat[x_src, x_src]
        new_src = mlp(sd, p + ".node_mlp", torch.cat([x_src, x_src], dim=1), act, n_extra_layers) + x_src
    return (new_src, new_dst), edges_new


def gnn_forward_mapper(sd: SD, p: str, x_src: Tensor, x_dst: Tensor, edge_attr_buf: Tensor, edge_index_base: Tensor,
                       batch_size: int, act: str = "SiLU", n_extra_layers: int = 0):
    """layers/mapper.py:525-608 + :485-522 + :108-116 ``GNNForwardMapper.forward``: returns (x_src_new, x_dst_new)."""
    edge_attr = trainable_tensor(edge_attr_buf, sd.get(p + ".trainable.trainable"), batch_size)
    edge_index = expand_edges(edge_index_base, sd[p + ".edge_inc"], batch_size)
    edge_attr = mlp(sd, p + ".emb_edges", edge_attr, act, n_extra_layers)
    xs = mlp(sd, p + ".emb_nodes_src", x_src, act, n_extra_layers)
    xd = mlp(sd, p + ".emb_nodes_dst", x_dst, act, n_extra_layers)
    (xs, xd), _ = gnn_mapper_block(sd, p + ".proc", xs, xd, edge_attr, edge_index, True, act, n_extra_layers)
    return xs, xd


def gnn_backward_mapper(sd: SD, p: str, x_src: Tensor, x_dst: Tensor, edge_attr_buf: Tensor, edge_index_base: Tensor,
                        batch_size: int, act: str = "SiLU", n_extra_layers: int = 0) -> Tensor:
    """layers/mapper.py:611-705 + :96-102 ``GNNBackwardMapper.forward``: no node embedding, MLP extractor without LN."""
    edge_attr = trainable_tensor(edge_attr_buf, sd.get(p + ".trainable.trainable"), batch_size)
    edge_index = expand_edges(edge_index_base, sd[p + ".edge_inc"], batch_size)
    edge_attr = mlp(sd, p + ".emb_edges", edge_attr, act, n_extra_layers)
    (_, xd), _ = gnn_mapper_block(sd, p + ".proc", x_src, x_dst, edge_attr, edge_index, False, act, n_extra_layers)
    return mlp(sd, p + ".node_data_extractor", xd, act, n_extra_layers, layer_norm=False)


# --------------------------------------------------------------------------
# Transformer (MHSA) path
# --------------------------------------------------------------------------


def mhsa(sd: SD, p: str, x: Tensor, batch_size: int, num_heads: int, window_size: Optional[int] = None) -> Tensor:
    """layers/attention.py:67-112 ``MultiHeadSelfAttention.forward``.

    ``window_size=None`` is the reference's SDPA fallback (global attention, the
    window is ignored there, attention.py:99-105).  With an integer window this
    follows flash-attn's ``window_size=(w, w)`` semantics (attention.py:96):
    key ``j`` is visible from query ``i`` iff ``|i - j| <= w``.
    """
    c = x.shape[1]
    d = c // num_heads
    qkv = _lin(sd, p + ".lin_qkv", x)
    q, k, v = qkv.chunk(3, -1)
    g = x.shape[0] // batch_size

    def heads(t):
        return t.reshape(batch_size, g, num_heads, d).permute(0, 2, 1, 3)

    q, k, v = heads(q), heads(k), heads(v)
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(d)
    if window_size is not None:
        i = torch.arange(g)
        mask = (i[:, None] - i[None, :]).abs() <= window_size
        s = s.masked_fill(~mask, float("-inf"))
    out = torch.matmul(torch.softmax(s, dim=-1), v)
    out = out.permute(0, 2, 1, 3).reshape(batch_size * g, c)
    return _lin(sd, p + ".projection", out)


def transformer_block(sd: SD, p: str, x: Tensor, batch_size: int, num_heads: int, act: str = "GELU",
                      window_size: Optional[int] = None) -> Tensor:
    """layers/block.py:99-105 ``TransformerProcessorBlock.forward``."""
    x = x + mhsa(sd, p + ".attention", _ln(sd, p + ".layer_norm1", x), batch_size, num_heads, window_size)
    h = _act(act)(_lin(sd, p + ".mlp.0", _ln(sd, p + ".layer_norm2", x)))
    return x + _lin(sd, p + ".mlp.2", h)


def transformer_processor(sd: SD, p: str, x: Tensor, batch_size: int, num_layers: int, num_chunks: int,
                          num_heads: int, act: str = "GELU", window_size: Optional[int] = None) -> Tensor:
    """layers/processor.py:145-162 + layers/chunk.py:108-114 ``TransformerProcessor.forward``."""
    per_chunk = num_layers // num_chunks
    for c in range(num_chunks):
        for b in range(per_chunk):
            x = transformer_block(sd, f"{p}.proc.{c}.blocks.{b}", x, batch_size, num_heads, act, window_size)
    return x


# --------------------------------------------------------------------------
# Model root
# --------------------------------------------------------------------------


def node_attributes(sd: SD, name: str, batch_size: int) -> Tensor:
    """layers/graph.py:107-113 ``NamedNodesAttributes.forward``."""
    return trainable_tensor(
        sd[f"node_attributes.latlons_{name}"],
        sd.get(f"node_attributes.trainable_tensors.{name}.trainable"),
        batch_size,
    )


def model_forward(
    sd: SD,
    graph: Mapping[str, Tensor],
    x: Tensor,
    *,
    num_heads: int,
    num_layers: int,
    num_chunks: int,
    prognostic_in: Sequence[int],
    prognostic_out: Sequence[int],
    processor: str = "GraphTransformer",
    act: str = "GELU",
    mapper_chunks: int = 1,
    data: str = "data",
    hidden: str = "hidden",
    gnn_act: str = "SiLU",
    window_size: Optional[int] = None,
    return_stages: bool = False,
    mappers: str = "GraphTransformer",
    boundings: Sequence[Mapping] = (),
    name_to_index_out: Optional[Mapping[str, int]] = None,
):
    """models/encoder_processor_decoder.py:168-233 ``AnemoiModelEncProcDec.forward``; ``boundings`` = the config's
    ``model.bounding`` list (dicts with ``_target_``), applied in order on the output (:229-231).

    ``graph`` holds ``enc_edge_index``/``enc_edge_attr`` (data->hidden),
    ``proc_edge_index``/``proc_edge_attr`` (hidden->hidden) and
    ``dec_edge_index``/``dec_edge_attr`` (hidden->data): the non-persistent
    buffers ``edge_index_base`` / ``edge_attr`` of layers/mapper.py:141-148.
    """
    b, t, ens, g, v = x.shape
    x_data = torch.cat(
        (x.permute(0, 2, 3, 1, 4).reshape(b * ens * g, t * v), node_attributes(sd, data, b)), dim=-1
    )  # :173-179
    x_hidden = node_attributes(sd, hidden, b)  # :181

    if mappers == "GNN":
        x_data_latent, x_latent = gnn_forward_mapper(sd, "encoder", x_data, x_hidden, graph["enc_edge_attr"],
                                                     graph["enc_edge_index"], b, gnn_act)
    else:
        x_data_latent, x_latent = gt_forward_mapper(
            sd, "encoder", x_data, x_hidden, graph["enc_edge_attr"], graph["enc_edge_index"], b, num_heads, act,
            mapper_chunks,
        )  # :188-194
    if processor == "GraphTransformer":
        x_proc = gt_processor(sd, "processor", x_latent, graph["proc_edge_attr"], graph["proc_edge_index"], b,
                              num_layers, num_chunks, num_heads, act)
    elif processor == "GNN":
        x_proc = gnn_processor(sd, "processor", x_latent, graph["proc_edge_attr"], graph["proc_edge_index"], b,
                               num_layers, num_chunks, gnn_act)
    elif processor == "Transformer":
        x_proc = transformer_processor(sd, "processor", x_latent, b, num_layers, num_chunks, num_heads, act,
                                       window_size)
    else:
        raise ValueError(processor)
    x_latent_proc = x_proc + x_latent  # :204
    if mappers == "GNN":
        x_out = gnn_backward_mapper(sd, "decoder", x_latent_proc, x_data_latent, graph["dec_edge_attr"],
                                    graph["dec_edge_index"], b, gnn_act)
    else:
        x_out = gt_backward_mapper(
            sd, "decoder", x_latent_proc, x_data_latent, graph["dec_edge_attr"], graph["dec_edge_index"], b,
            num_heads, act, mapper_chunks,
        )  # :207-213
    x_out = x_out.reshape(b, ens, g, -1).to(x.dtype).clone()  # :215-224
    x_out[..., list(prognostic_out)] += x[:, -1, :, :, list(prognostic_in)]  # :227
    if boundings:
        x_out = apply_boundings(x_out, boundings, name_to_index_out)  # :229-231
    if return_stages:
        return x_out, {"x_latent": x_latent, "x_proc": x_proc}
    return x_out


def hierarchical_forward(
    sd: SD,
    graph: Mapping[str, Tensor],
    x: Tensor,
    *,
    hidden: Sequence[str],
    num_heads: int,
    level_layers: int,
    prognostic_in: Sequence[int],
    prognostic_out: Sequence[int],
    level_process: bool = True,
    act: str = "GELU",
    data: str = "data",
):
    """models/hierarchical.py:178-308 ``AnemoiModelEncProcDecHierarchical.forward`` with GraphTransformer mappers /
    processors (no boundings).

    ``graph`` holds, per sub-module prefix ``p`` (``encoder``, ``decoder``, ``downscale.<h>``, ``upscale.<h>``,
    ``down_level_processor.<h>``, ``up_level_processor.<h>``), the buffers ``p + ".edge_index"`` and
    ``p + ".edge_attr"`` (``edge_index_base`` / ``edge_attr`` of layers/mapper.py:141-148).
    """
    b, t, ens, g, v = x.shape
    x_data = torch.cat(
        (x.permute(0, 2, 3, 1, 4).reshape(b * ens * g, t * v), node_attributes(sd, data, b)), dim=-1
    )  # :183-190
    x_hid = {h: node_attributes(sd, h, b) for h in hidden}  # :193-195

    def fwd(p, xs, xd):
        return gt_forward_mapper(sd, p, xs, xd, graph[p + ".edge_attr"], graph[p + ".edge_index"], b, num_heads, act)

    def bwd(p, xs, xd):
        return gt_backward_mapper(sd, p, xs, xd, graph[p + ".edge_attr"], graph[p + ".edge_index"], b, num_heads, act)

    def proc(p, xx):
        return gt_processor(sd, p, xx, graph[p + ".edge_attr"], graph[p + ".edge_index"], b, level_layers, 1, num_heads,
                            act)

    x_data_latent, curr = fwd("encoder", x_data, x_hid[hidden[0]])  # :204-210
    enc, skip = {}, {}
    for src, dst in zip(hidden[:-1], hidden[1:]):  # :217-241
        if level_process:
            curr = proc(f"down_level_processor.{src}", curr)
        skip[src] = curr
        enc[src], curr = fwd(f"downscale.{src}", curr, x_hid[dst])
    if level_process:  # :244-250
        curr = proc(f"down_level_processor.{hidden[-1]}", curr)
    for dst, src in zip(reversed(hidden[:-1]), reversed(hidden[1:])):  # :253-278
        curr = bwd(f"upscale.{src}", curr, enc[dst])
        curr = curr + skip[dst]
        if level_process:
            curr = proc(f"up_level_processor.{dst}", curr)
    x_out = bwd("decoder", curr, x_data_latent)  # :281-287
    x_out = x_out.reshape(b, ens, g, -1).to(x.dtype).clone()  # :289-299
    x_out[..., list(prognostic_out)] += x[:, -1, :, :, list(prognostic_in)]  # :302
    return x_out


# --------------------------------------------------------------------------
# Interface: normaliser + predict_step (the step either side of the model forward)
# --------------------------------------------------------------------------
def apply_boundings(x: Tensor, boundings: Sequence[Mapping], name_to_index: Mapping[str, int]) -> Tensor:
    """layers/bounding.py:60-124 ``ReluBounding`` / ``HardtanhBounding`` / ``FractionBounding`` chained in config order,
    in place; the columns are the SORTED indices of ``variables`` (data_indices/tensor.py:91-94)."""
    for cfg in boundings:
        kind = cfg["_target_"].rsplit(".", 1)[-1]
        idx = sorted(name_to_index[v] for v in cfg["variables"])
        if kind == "ReluBounding":  # :60-65
            x[..., idx] = torch.nn.functional.relu(x[..., idx])
        elif kind in ("HardtanhBounding", "FractionBounding"):  # :68-92
            x[..., idx] = torch.nn.functional.hardtanh(x[..., idx], min_val=cfg["min_val"], max_val=cfg["max_val"])
            if kind == "FractionBounding":  # :95-124
                x[..., idx] *= x[..., [name_to_index[cfg["total_var"]]]]
        else:
            raise ValueError(kind)
    return x


def normalizer_affine(config: Mapping, name_to_index: Mapping[str, int], statistics: Mapping):
    """preprocessing/normalizer.py:44-105 (+ preprocessing/__init__.py:63-101): per-variable ``(mul, add)`` of
    ``InputNormalizer`` from its config ``{default, remap, <method>: [variables]}`` and the dataset statistics."""
    import numpy as np

    default = config.get("default", "none")
    remap = config.get("remap", {})
    methods = {}
    for method, variables in config.items():
        if method in ("default", "remap") or variables is None or variables == "none":
            continue
        for v in ([variables] if isinstance(variables, str) else variables):
            methods[v] = method
    mn, mx = np.array(statistics["minimum"], copy=True), np.array(statistics["maximum"], copy=True)
    mean, sd = np.array(statistics["mean"], copy=True), np.array(statistics["stdev"], copy=True)
    new = {name_to_index[r]: (mn[name_to_index[s_]], mx[name_to_index[s_]], mean[name_to_index[s_]], sd[name_to_index[s_]])
           for r, s_ in remap.items()}  # :53-60 two-step remap
    for i, (a, b, c, d) in new.items():
        mn[i], mx[i], mean[i], sd[i] = a, b, c, d
    mul = np.ones((mn.size,), dtype=np.float32)
    add = np.zeros((mn.size,), dtype=np.float32)
    for name, i in name_to_index.items():
        m = methods.get(name, default)
        if m == "mean-std":  # :70-75
            mul[i], add[i] = 1 / sd[i], -mean[i] / sd[i]
        elif m == "std":  # :77-82
            mul[i], add[i] = 1 / sd[i], 0
        elif m == "min-max":  # :84-90
            mul[i], add[i] = 1 / (mx[i] - mn[i]), -mn[i] / (mx[i] - mn[i])
        elif m == "max":  # :92-94
            mul[i] = 1 / mx[i]
        elif m != "none":
            raise ValueError(m)
    return torch.from_numpy(mul), torch.from_numpy(add)


def predict_step(sd: SD, graph: Mapping[str, Tensor], batch: Tensor, *, multi_step: int, **model_kwargs) -> Tensor:
    """interface/__init__.py:97-123 ``AnemoiModelInterface.predict_step`` with one ``InputNormalizer`` named
    ``normalizer``: normalise the input variables (preprocessing/normalizer.py:134-164, branch
    ``x.shape[-1] == len(_input_idx)``), run the model, de-normalise the output variables (:166-205)."""
    p = "pre_processors.processors.normalizer."
    mul, add = sd[p + "_norm_mul"], sd[p + "_norm_add"]
    i_in, i_out = sd[p + "_input_idx"].long(), sd[p + "_output_idx"].long()
    x = batch * mul[i_in] + add[i_in]
    x = x[:, 0:multi_step, None, ...]
    model_sd = {k[len("model."):]: v for k, v in sd.items() if k.startswith("model.")}
    y = model_forward(model_sd, graph, x, **model_kwargs)
    return (y - add[i_out]) / mul[i_out]


def rollout(sd: SD, graph: Mapping[str, Tensor], batch: Tensor, n_steps: int, forcings: Optional[Tensor] = None, *,
            multi_step: int, prognostic_in, prognostic_out, forcing_in=(), **model_kwargs) -> Tensor:
    """``n_steps`` autoregressive applications of :func:`predict_step`'s model (BASELINE config 4).

    PARITY UNPINNED against the reference for the loop itself: the reference repository stops at one step
    (interface/__init__.py:97-123); the loop is its caller's (anemoi-training ``GraphForecaster.advance_input``): roll
    the time axis of the NORMALISED input by one, write the prognostic outputs of the prediction into the last slice,
    write the (normalised) forcings of the new time there, keep every other variable.  Each step's arithmetic is the
    pinned :func:`model_forward`.  ``batch`` is ``[B, T, G, V_in]`` physical, ``forcings`` ``[n_steps, B, G, F]``
    physical values of the forcing inputs valid at the time of step ``s``'s OUTPUT (consumed by step ``s + 1``);
    returns ``[n_steps, B, 1, G, V_out]`` physical predictions.
    """
    p = "pre_processors.processors.normalizer."
    mul, add = sd[p + "_norm_mul"], sd[p + "_norm_add"]
    i_in, i_out = sd[p + "_input_idx"].long(), sd[p + "_output_idx"].long()
    model_sd = {k[len("model."):]: v for k, v in sd.items() if k.startswith("model.")}
    pin, pout, fin = list(prognostic_in), list(prognostic_out), list(forcing_in)
    x = (batch * mul[i_in] + add[i_in])[:, 0:multi_step, None, ...].clone()
    outs = []
    for s in range(n_steps):
        y = model_forward(model_sd, graph, x, prognostic_in=pin, prognostic_out=pout, **model_kwargs)
        outs.append((y - add[i_out]) / mul[i_out])
        nxt = x.roll(-1, dims=1)
        nxt[:, -1] = x[:, -1]
        nxt[:, -1, :, :, pin] = y[..., pout]
        if forcings is not None and fin:
            f = forcings[s] * mul[i_in][fin] + add[i_in][fin]
            nxt[:, -1, :, :, fin] = f[:, None]
        x = nxt
    return torch.stack(outs)
