"""Differentiable forwards of the module tree: what ``module(x)`` runs when autograd is on.

Every ``nn.Module`` of this package keeps two routes to the same arithmetic:

* ``native`` -- the inference launch sequence (packed weights, LayerNorm folded into GEMMs, fused epilogues, HIP graph
  friendly); taken under ``torch.no_grad()``;
* the functions below -- the same layers composed from the ``torch.autograd.Function``s of :mod:`anemoi_models_amd.autograd`
  (fused Linear, LayerNorm, the folded graph-transformer edge phase, the GNN gather / segment-sum), each with its forward
  AND backward on the HIP kernels.  ``module.forward`` dispatches here whenever a gradient is wanted
  (:func:`wants_grad`), so ``loss.backward()`` works on a block, a mapper, a processor or a whole model exactly as in
  the reference's own tests (reference tests/layers/processor/test_graphtransformer_processor.py:145-159) and in
  anemoi-training.

Model families: flat and hierarchical GraphTransformer models, the GNN processor and GNN mappers, the Transformer
processor (bf16 head sizes 64 / 32: MFMA attention forward and backward; other cases and attention dropout: VALU kernels).

Activation checkpointing follows the reference: every mapper call and every processor chunk is wrapped in
``torch.utils.checkpoint`` (reference models/encoder_processor_decoder.py:159-166, layers/processor.py:73-77); all
kernels are deterministic, so the recomputation reproduces the forward bit for bit.  The decoder -- the last region, whose
recomputation could free nothing at the memory peak -- keeps its activations (:func:`_checkpoint`).  ``ANEMOI_AMD_CHECKPOINT=0``
keeps every activation instead (faster, config 3 then needs ~58 GiB per sample).
"""

from __future__ import annotations

import os
from typing import Optional

import torch
from torch import Tensor
from torch import nn
from torch.utils.checkpoint import checkpoint as _torch_checkpoint

from . import autograd
from . import ops
from . import runtime


def wants_grad(module: nn.Module, *tensors) -> bool:
    """True when this call must build an autograd graph: grad mode on and a parameter or an input asks for a gradient."""
    if not torch.is_grad_enabled():
        return False
    if any(isinstance(t, Tensor) and t.requires_grad for t in tensors):
        return True
    return any(p.requires_grad for p in module.parameters())


def _checkpoint(fn, *args, last: bool = False):
    """``last``: the region is the LAST one of the forward (the decoder).  Its recomputation would run at the very start of
    the backward, while every other region's checkpoint is still held: it frees nothing at the peak and costs a second
    decoder forward (13 ms of a 153 ms config-3 step) -- so it keeps its activations instead
    (``ANEMOI_AMD_CHECKPOINT_LAST=1`` wraps it like the reference does, models/encoder_processor_decoder.py:223-231)."""
    if os.environ.get("ANEMOI_AMD_CHECKPOINT", "1") == "0":
        return fn(*args)
    if last and os.environ.get("ANEMOI_AMD_CHECKPOINT_LAST", "0") != "1":
        return fn(*args)
    dd = runtime.device_dropout()
    if dd is None:  # host-drawn dropout seeds: torch's checkpoint restores the CPU generator for the recomputation
        return _torch_checkpoint(fn, *args, use_reentrant=False)
    # device-side dropout seeds: the recomputation has to see the step word of ITS forward, wherever and whenever the
    # backward runs (outside the ``with DeviceDropout`` block the modules would draw fresh host seeds, after a later
    # ``advance()`` the next step's word: either way another mask than the forward's, i.e. silently wrong gradients)
    frozen = dd.pinned()

    def pinned(*a):
        with frozen:
            return fn(*a)

    return _torch_checkpoint(pinned, *args, use_reentrant=False)


def _cast(x: Tensor, dtype: torch.dtype) -> Tensor:
    x = x if x.dtype == dtype else x.to(dtype)
    return x if x.stride(-1) == 1 else x.contiguous()


# ------------------------------------------------------------------------------------------------ MLP / Sequential
def sequential(seq: nn.Sequential, x: Tensor, residual: Optional[Tensor] = None) -> Tensor:
    """Linear / activation / LayerNorm stack with each Linear fused with the activation behind it."""
    from .layers.mlp import fused_activation_name

    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if (isinstance(m, nn.Linear) and i + 2 < len(mods) and isinstance(mods[i + 2], nn.Linear)
                and not isinstance(mods[i + 1], (nn.Linear, nn.LayerNorm)) and fused_activation_name(mods[i + 1])
                and (i + 3 == len(mods) or isinstance(mods[i + 3], (nn.Linear, nn.LayerNorm)))):
            # Linear -> activation -> Linear: one autograd node (pre-activation from the first GEMM's epilogue, act' in the
            # second GEMM's backward epilogue)
            last = i + 2 == len(mods) - 1
            x = autograd.mlp2(x, m.weight, m.bias, mods[i + 2].weight, mods[i + 2].bias,
                              fused_activation_name(mods[i + 1]), residual if last else None)
            if last:
                residual = None
            i += 3
            continue
        if isinstance(m, nn.Linear):
            act = "Identity"
            if (i + 1 < len(mods) and not isinstance(mods[i + 1], (nn.Linear, nn.LayerNorm))
                    and fused_activation_name(mods[i + 1])):
                act = fused_activation_name(mods[i + 1])
                i += 1
            last = i == len(mods) - 1
            x = autograd.linear(x, m.weight, m.bias, act, residual if last else None)
            if last:
                residual = None
        elif isinstance(m, nn.LayerNorm):
            x = autograd.layer_norm(x, m.weight, m.bias, m.eps)
        else:  # an activation without a GEMM epilogue (the reference takes any torch.nn activation): plain torch autograd
            x = m(x)
        i += 1
    return x if residual is None else x + residual


def mlp(module, x: Tensor, residual: Optional[Tensor] = None) -> Tensor:
    """``layers.mlp.MLP`` (reference layers/mlp.py:74-89)."""
    from .layers.utils import CheckpointWrapper

    seq = module.model.module if isinstance(module.model, CheckpointWrapper) else module.model
    return sequential(seq, x, residual)


# ------------------------------------------------------------------------------------------------ graph transformer
def _block_sd(block: nn.Module) -> dict:
    return {"b." + k: v for k, v in block.named_parameters()}


def _check_heads(channels: int, num_heads: int, dtype: torch.dtype) -> None:
    # (any head size the conv kernels reach: the folded kernels where they exist for it, else lin_edge as a GEMM and the
    #  conv on explicit edge features with zero-padded heads -- autograd.folded_edge_route / conv_head_size; until round 6
    #  training needed a head size that is a multiple of 4)
    if channels % num_heads != 0:
        raise ValueError(f"{channels} channels do not split into {num_heads} heads")


def _gt_edge_inputs(block, edge_attr: Tensor, edge_index: Tensor, n_src: int, n_dst: int):
    plan = block._plans.get(edge_index, n_src, n_dst)
    up = ops.round_up(edge_attr.shape[1] + 1, 4)
    return plan, autograd._edge_attr_csr(edge_attr, None, plan, up)


def gt_processor_block(block, x: Tensor, edge_attr: Tensor, edge_index: Tensor, size=None):
    """``GraphTransformerProcessorBlock.forward`` (reference layers/block.py:602-635) with an autograd graph."""
    dtype = runtime.compute_dtype(x)
    n = x.shape[0]
    if size is not None and tuple(size) != (n, n):
        raise ValueError(f"Encountered tensor with size {n} in dimension 0, but expected size {tuple(size)}")
    _check_heads(x.shape[1], block.num_heads, dtype)
    plan, ea = _gt_edge_inputs(block, edge_attr, edge_index, n, n)
    y = autograd.gt_processor_block(_cast(x, dtype), _block_sd(block), "b", ea, plan, block.num_heads, block.activation,
                                    block.layer_norm1.eps)
    return y, edge_attr


def gt_mapper_block(block, x, edge_attr: Tensor, edge_index: Tensor, size=None):
    """``GraphTransformerMapperBlock.forward`` (reference layers/block.py:479-550)."""
    x_src, x_dst = x
    dtype = runtime.compute_dtype(x_dst)
    n_src, n_dst = x_src.shape[0], x_dst.shape[0]
    if size is not None and tuple(size) != (n_src, n_dst):
        raise ValueError(f"Encountered tensors with sizes {(n_src, n_dst)}, but expected size {tuple(size)}")
    _check_heads(x_dst.shape[1], block.num_heads, dtype)
    plan, ea = _gt_edge_inputs(block, edge_attr, edge_index, n_src, n_dst)
    y = autograd.gt_mapper_block(_cast(x_src, dtype), _cast(x_dst, dtype), _block_sd(block), "b", ea, plan, block.num_heads,
                                 block.activation, block.layer_norm1.eps)
    if block.update_src_nodes:  # row-local MLP of the source rows (reference layers/block.py:540-546)
        xs = _cast(x_src, dtype)
        return (sequential(block.node_src_mlp, xs, residual=xs), y), edge_attr
    return (x_src, y), edge_attr


def _set_plan_and_attrs(mod, n_src: int, n_dst: int, batch_size: int, up: Optional[int] = None,
                        src_map: Optional[Tensor] = None, dst_map: Optional[Tensor] = None):
    """(plan, edge attributes in CSR order) of a mapper / processor: ``cat[edge_attr, trainable]`` repeated per batch
    element (reference layers/graph.py:37-44) on the batched graph (layers/mapper.py:150-171).  ``src_map`` / ``dst_map``:
    external node id -> row, when the caller keeps a node set in an internal order (the model root: Morton-ordered mesh)."""
    plan = mod._plans.get(mod.edge_index_base, n_src, n_dst, batch_size, mod.edge_inc, src_map, dst_map)
    trainable = mod.trainable.trainable
    if up is None:  # GNN: the plain attribute matrix
        parts = [mod.edge_attr.float()] + ([] if trainable is None else [trainable.float()])
        return plan, autograd.permute_rows(torch.cat(parts, dim=1).repeat(batch_size, 1), plan.perm.long())
    return plan, autograd._edge_attr_csr(mod.edge_attr, trainable, plan, up, batch_size)


def gt_processor(proc, x: Tensor, batch_size: int, node_map: Optional[Tensor] = None) -> Tensor:
    """``GraphTransformerProcessor.forward`` (reference layers/processor.py:317-343): checkpointed chunks of blocks that
    share ONE plan and ONE CSR attribute matrix."""
    dtype = runtime.compute_dtype(x)
    blk0 = proc.proc[0].blocks[0]
    _check_heads(x.shape[1], blk0.num_heads, dtype)
    n = x.shape[0]
    plan, ea = _set_plan_and_attrs(proc, n, n, batch_size, ops.round_up(proc.edge_dim + 1, 4), node_map, node_map)

    # what the blocks derive from their parameters alone (lin_edge fold, weight assembly, casts, transposes): once for
    # all blocks of the processor, outside the checkpointed chunks
    blocks = [blk for chunk in proc.proc for blk in chunk.blocks]
    sds = [_block_sd(blk) for blk in blocks]
    same = all(b.num_heads == blk0.num_heads and b.activation == blk0.activation for b in blocks)
    prepared = autograd.gt_processor_weights(sds, "b", x.shape[1], blk0.num_heads, ea.shape[1], dtype, x.device) if same else None
    index = {id(blk): i for i, blk in enumerate(blocks)}

    def run_chunk(chunk, h, attrs):
        for blk in chunk.blocks:
            i = index[id(blk)]
            h = autograd.gt_processor_block(h, sds[i], "b", attrs, plan, blk.num_heads, blk.activation,
                                            blk.layer_norm1.eps, None if prepared is None else prepared[i])
        return h

    h = _cast(x, dtype)
    for chunk in proc.proc:
        h = _checkpoint(run_chunk, chunk, h, ea)
    return h


def gt_mapper(mapper, x_src: Tensor, x_dst: Tensor, batch_size: int, src_map: Optional[Tensor] = None,
              dst_map: Optional[Tensor] = None, augmented: bool = False) -> Tensor:
    """``GraphTransformerForwardMapper`` / ``GraphTransformerBackwardMapper`` (reference layers/mapper.py:275-418): returns
    the mapped (and, for the backward mapper, extracted) destination nodes.  ``augmented``: the rows of the LARGER node set (the
    grid) are ``[features | 1 | 0-pad]`` already (:class:`_AssembleNodes`)."""
    dtype = runtime.compute_dtype(x_dst)
    blk = mapper.proc
    _check_heads(mapper.hidden_dim, blk.num_heads, dtype)
    plan, ea = _set_plan_and_attrs(mapper, x_src.shape[0], x_dst.shape[0], batch_size, ops.round_up(mapper.edge_dim + 1, 4),
                                   src_map, dst_map)
    hs, hd = _cast(x_src, dtype), _cast(x_dst, dtype)
    # the LARGER node set (the grid: sources of the encoder, destinations of the decoder) feeds its first Linear from the
    # raw features -- embedding -> LayerNorm -> Linear folded into a K = features + 1 product (bf16, few input features:
    # the inference path's rule, layers/mapper.py::_embedded); the decoder still forms its embedded rows (skip connection)
    kv_fn = sq_fn = None
    eps = blk.layer_norm1.eps

    def can_fold(lin):
        return (runtime.embed_fold_enabled(dtype) and 2 * (lin.in_features + 1) <= lin.out_features
                and autograd.FOLD_MAX_UP >= ops.round_up(mapper.edge_dim + 1, 4))

    if hasattr(mapper, "emb_nodes_src") and hs.shape[0] >= hd.shape[0] and can_fold(mapper.emb_nodes_src):
        raw_src, emb = hs, mapper.emb_nodes_src
        kv_fn = lambda w, b, gamma, beta: autograd.folded_embedding_ln_linear(  # noqa: E731
            raw_src, emb.weight, emb.bias, gamma, beta, eps, w, b, augmented=augmented)
        hs = None
    elif hasattr(mapper, "emb_nodes_src"):
        hs = autograd.linear(hs, mapper.emb_nodes_src.weight, mapper.emb_nodes_src.bias, padded_input=augmented)
    if hd.shape[0] > (0 if hs is None else hs.shape[0]) and hs is not None and can_fold(mapper.emb_nodes_dst):
        raw_dst, emb_d = hd, mapper.emb_nodes_dst
        sq_fn = lambda w, b, gamma, beta: autograd.folded_embedding_ln_linear(  # noqa: E731
            raw_dst, emb_d.weight, emb_d.bias, gamma, beta, eps, w, b, augmented=augmented)
    hd = autograd.linear(hd, mapper.emb_nodes_dst.weight, mapper.emb_nodes_dst.bias, padded_input=augmented)
    y = autograd.gt_mapper_block(hs, hd, _block_sd(blk), "b", ea, plan, blk.num_heads, blk.activation, eps,
                                 kv_fn=kv_fn, sq_fn=sq_fn)
    ext = getattr(mapper, "node_data_extractor", None)
    if ext is not None:
        y = sequential(ext, y)
    return y


# ------------------------------------------------------------------------------------------------ GNN
def _gnn_edge_update(conv_mlp, x_dst: Tensor, x_src: Tensor, e_csr: Tensor, plan) -> Tensor:
    """``edge_mlp(cat[x_i, x_j, e]) + e`` (reference layers/conv.py:47-76) without the ``[E, 3C]`` concatenation: the
    first Linear splits into a destination part, a source part (node GEMMs) and an edge part (edge GEMM); the sum and the
    activation are the gather kernel."""
    from .layers.utils import CheckpointWrapper

    seq = conv_mlp.model.module if isinstance(conv_mlp.model, CheckpointWrapper) else conv_mlp.model
    mods = list(seq)
    lin1 = mods[0]
    c = x_dst.shape[1]
    if not isinstance(lin1, nn.Linear) or lin1.in_features != 3 * c or e_csr.shape[1] != c:
        raise ValueError(f"GNN block expects node / edge width {lin1.in_features // 3}, got {c} / {e_csr.shape[1]}")
    act1, rest = "Identity", mods[1:]
    if rest and not isinstance(rest[0], (nn.Linear, nn.LayerNorm)):
        act1, rest = type(rest[0]).__name__, rest[1:]
    p_dst = autograd.linear(x_dst, lin1.weight[:, :c], None)
    p_src = autograd.linear(x_src, lin1.weight[:, c:2 * c], None)
    t = autograd.linear(e_csr, lin1.weight[:, 2 * c:], lin1.bias)
    h = autograd.gather_add_act(t, p_dst, p_src, plan, act1)
    return sequential(nn.Sequential(*rest), h, residual=e_csr)


def gnn_message_pass(conv_mlp, x_dst: Tensor, x_src: Tensor, e_csr: Tensor, plan):
    """``(new edge state, its sum over the destinations)`` of ``GraphConv`` (reference layers/conv.py:62-76).  An edge set
    without edges (PyG's propagate takes one) has an empty new state and zero sums: nothing is launched on empty operands."""
    if e_csr.shape[0] == 0:
        return e_csr, x_dst.new_zeros(x_dst.shape)
    e_new = _gnn_edge_update(conv_mlp, x_dst, x_src, e_csr, plan)
    return e_new, autograd.segment_sum(e_new, plan)


def gnn_processor_block_csr(block, x: Tensor, e_csr: Tensor, plan):
    """``GraphConvProcessorBlock`` (reference layers/block.py:193-223) on an edge state kept in CSR order."""
    e_new, agg = gnn_message_pass(block.conv.edge_mlp, x, x, e_csr, plan)
    return mlp(block.node_mlp, torch.cat([x, agg], dim=1), residual=x), e_new


def gnn_mapper_block_csr(block, x_src: Tensor, x_dst: Tensor, e_csr: Tensor, plan):
    """``GraphConvMapperBlock`` (reference layers/block.py:226-286)."""
    e_new, agg = gnn_message_pass(block.conv.edge_mlp, x_dst, x_src, e_csr, plan)
    new_dst = mlp(block.node_mlp, torch.cat([x_dst, agg], dim=1), residual=x_dst)
    new_src = mlp(block.node_mlp, torch.cat([x_src, x_src], dim=1), residual=x_src) if block.update_src_nodes else x_src
    return (new_src, new_dst), e_new


def _csr_round_trip(plan, edge_attr: Tensor, dtype):
    perm = plan.perm.long()
    if perm.numel() == 0:
        return _cast(edge_attr, dtype), perm
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(perm.shape[0], device=perm.device)
    return autograd.permute_rows(_cast(edge_attr, dtype), perm), inv


def gnn_processor_block(block, x: Tensor, edge_attr: Tensor, edge_index: Tensor, size=None):
    """``GraphConvProcessorBlock.forward`` with the caller's edge order in and out."""
    dtype = runtime.compute_dtype(x)
    n = x.shape[0]
    plan = block._plans.get(edge_index, n, n)
    e_csr, inv = _csr_round_trip(plan, edge_attr, dtype)
    x_new, e_new = gnn_processor_block_csr(block, _cast(x, dtype), e_csr, plan)
    return x_new, autograd.permute_rows(e_new, inv)


def gnn_mapper_block(block, x, edge_attr: Tensor, edge_index: Tensor, size=None):
    x_src, x_dst = x
    dtype = runtime.compute_dtype(x_dst)
    n_src, n_dst = x_src.shape[0], x_dst.shape[0]
    if size is not None and tuple(size) != (n_src, n_dst):
        raise ValueError(f"Encountered tensors with sizes {(n_src, n_dst)}, but expected size {tuple(size)}")
    plan = block._plans.get(edge_index, n_src, n_dst)
    e_csr, inv = _csr_round_trip(plan, edge_attr, dtype)
    nodes, e_new = gnn_mapper_block_csr(block, _cast(x_src, dtype), _cast(x_dst, dtype), e_csr, plan)
    return nodes, autograd.permute_rows(e_new, inv)


def gnn_processor(proc, x: Tensor, batch_size: int, node_map: Optional[Tensor] = None) -> Tensor:
    """``GNNProcessor.forward`` (reference layers/processor.py:228-250, layers/chunk.py:165-181)."""
    dtype = runtime.compute_dtype(x)
    n = x.shape[0]
    plan, ea = _set_plan_and_attrs(proc, n, n, batch_size, None, node_map, node_map)

    def run_chunk(chunk, h, e):
        if chunk.emb_edges is not None:
            e = mlp(chunk.emb_edges, e)
        for blk in chunk.blocks:
            h, e = gnn_processor_block_csr(blk, h, e, plan)
        return h, e

    h, e = _cast(x, dtype), _cast(ea, dtype)
    for chunk in proc.proc:
        h, e = _checkpoint(run_chunk, chunk, h, e)
    return h


def gnn_mapper(mapper, x_src: Tensor, x_dst: Tensor, batch_size: int, src_map: Optional[Tensor] = None,
               dst_map: Optional[Tensor] = None):
    """``GNNForwardMapper`` / ``GNNBackwardMapper`` (reference layers/mapper.py:485-522, 600-705): returns
    ``(source nodes after the block, destination nodes after the block / extraction)``."""
    dtype = runtime.compute_dtype(x_dst)
    plan, ea = _set_plan_and_attrs(mapper, x_src.shape[0], x_dst.shape[0], batch_size, None, src_map, dst_map)
    e = mlp(mapper.emb_edges, _cast(ea, dtype))
    hs, hd = _cast(x_src, dtype), _cast(x_dst, dtype)
    if hasattr(mapper, "emb_nodes_src"):
        hs, hd = mlp(mapper.emb_nodes_src, hs), mlp(mapper.emb_nodes_dst, hd)
    (hs, hd), _ = gnn_mapper_block_csr(mapper.proc, hs, hd, e, plan)
    ext = getattr(mapper, "node_data_extractor", None)
    if ext is not None:
        hd = mlp(ext, hd)
    return hs, hd


# ------------------------------------------------------------------------------------------------ Transformer
def transformer_block(block, x: Tensor, batch_size: int) -> Tensor:
    """``TransformerProcessorBlock.forward`` (reference layers/block.py:99-105): ``x + proj(attn(qkv(LN x)))``, then
    ``x + MLP(LN x)``."""
    dtype = runtime.compute_dtype(x)
    att = block.attention
    x = _cast(x, dtype)
    h = autograd.layer_norm(x, block.layer_norm1.weight, block.layer_norm1.bias, block.layer_norm1.eps)
    qkv = autograd.linear(h, att.lin_qkv.weight, att.lin_qkv.bias)
    p, seed, seed_dev = att.dropout()  # attention dropout in training mode (reference layers/attention.py:90)
    a = autograd.mhsa(qkv, batch_size, att.num_heads, att.attention_window(), p, seed, seed_dev=seed_dev)
    x = autograd.linear(a, att.projection.weight, att.projection.bias, "Identity", x)
    h = autograd.layer_norm(x, block.layer_norm2.weight, block.layer_norm2.bias, block.layer_norm2.eps)
    return sequential(block.mlp, h, residual=x)


def transformer_processor(proc, x: Tensor, batch_size: int) -> Tensor:
    """``TransformerProcessor.forward`` (reference layers/processor.py:103-137): checkpointed chunks of blocks."""
    def run_chunk(chunk, h):
        for blk in chunk.blocks:
            h = transformer_block(blk, h, batch_size)
        return h

    h = _cast(x, runtime.compute_dtype(x))
    for chunk in proc.proc:
        h = _checkpoint(run_chunk, chunk, h)
    return h


# ------------------------------------------------------------------------------------------------ model roots
def _node_rows(model, name: str, rows: int) -> Tensor:
    na = model.node_attributes
    parts = [na.latlons(name)]
    tr = na.trainable_tensors[name].trainable
    if tr is not None:
        parts.append(tr)
    return torch.cat(parts, dim=1).repeat(rows, 1)


class _AssembleNodes(torch.autograd.Function):
    """The model input as the mappers' GEMMs read it -- rows ``(b, ens, g)`` of ``[x (time-major) | sin/cos latlon | trainable |
    1 | 0-pad]`` in the compute dtype, ``ld`` columns -- from ONE kernel (``anemoi_assemble_nodes``, the inference route's)
    instead of permute / cat / cast here and one more concatenation per folded embedding (four passes over the 542 080 grid rows
    of config 3).  The constant 1 behind the features carries the embedding bias of the folded products
    (``autograd.folded_embedding_ln_linear(augmented=True)``) and meets zero weights everywhere else.  Backward: the columns of
    the trainable tensor, summed over the batch; ``x`` carries no gradient on this route (the caller checks)."""

    @staticmethod
    def forward(ctx, x: Tensor, latlons: Tensor, trainable: Optional[Tensor], dtype: torch.dtype, ld: int):
        g = latlons.shape[0]
        ones = torch.ones((g, 1), dtype=torch.float32, device=latlons.device)
        tr1 = ones if trainable is None else torch.cat([trainable.detach().float(), ones], dim=1)
        out = ops.assemble_nodes(x, latlons, tr1, x.shape[0], dtype, ld_out=ld)
        ctx.off = x.shape[1] * x.shape[4] + latlons.shape[1]
        ctx.n_tr = 0 if trainable is None else trainable.shape[1]
        ctx.g, ctx.tr_dtype = g, (None if trainable is None else trainable.dtype)
        return out

    @staticmethod
    def backward(ctx, grad: Tensor):
        if ctx.n_tr == 0 or not ctx.needs_input_grad[2]:
            return None, None, None, None, None
        gt = grad[:, ctx.off:ctx.off + ctx.n_tr].float().reshape(-1, ctx.g, ctx.n_tr).sum(0)
        return None, None, gt.to(ctx.tr_dtype), None, None


class _PrognosticResidual(torch.autograd.Function):
    """``y = float(out)`` with ``y[..., prognostic] += x[:, -1, ..., prognostic_in]`` (reference
    models/encoder_processor_decoder.py:223-228) in the ONE pass of ``anemoi_finalize_output`` -- the inference route's
    kernel -- instead of zeros / index_select / index_put / add over the ``[grid, V_out]`` output (five passes over 173 MB at
    config 3).  ``x`` carries no gradient on this route (the caller checks); the gradient of ``out`` is the incoming one."""

    @staticmethod
    def forward(ctx, out: Tensor, x: Tensor, src: Tensor, shape):
        y = torch.empty(shape, dtype=torch.float32, device=out.device)  # (a tensor of its own, not a view: boundings write in place)
        y.view(out.shape).copy_(out)
        ops.finalize_output(y, x, src, None, None)
        ctx.out_shape, ctx.out_dtype = out.shape, out.dtype
        return y

    @staticmethod
    def backward(ctx, g: Tensor):
        return g.reshape(ctx.out_shape).to(ctx.out_dtype), None, None, None


def _finish(model, out: Tensor, x: Tensor, b: int, ens: int, g: int) -> Tensor:
    if (out.is_cuda and not x.requires_grad and x.dtype == torch.float32 and x.dim() == 5
            and os.environ.get("ANEMOI_AMD_TRAIN_FUSED_FINISH", "1") != "0"):
        key = ("residual_src", str(x.device))  # (the inference route's column map, models/encoder_processor_decoder.py::_finish)
        if key not in model._idx_cache:
            src = torch.full((model.num_output_channels,), -1, dtype=torch.int32)
            src[torch.as_tensor(model._internal_output_idx).long()] = torch.as_tensor(model._internal_input_idx).to(torch.int32)
            model._idx_cache[key] = src.to(x.device)
        y = _PrognosticResidual.apply(out, x, model._idx_cache[key], (b, ens, g, out.shape[-1]))
        for bounding in model.boundings:
            y = bounding(y)
        return y
    y = out.float().reshape(b, ens, g, -1).to(x.dtype).clone()
    key = ("prognostic_long", str(x.device))  # index tensors resident on the device: a Python list here costs an upload
    if key not in model._idx_cache:           # and a device synchronisation per step
        model._idx_cache[key] = tuple(torch.as_tensor(i).to(device=x.device, dtype=torch.int64)
                                      for i in (model._internal_output_idx, model._internal_input_idx))
    o_idx, i_idx = model._idx_cache[key]
    # the prognostic residual as ONE full-width addend (zeros outside the prognostic columns): a plain add in the graph --
    # reading y[..., o_idx] instead costs torch's sort-based index accumulation in the backward (1.8 ms per step)
    res = torch.zeros_like(y)
    res[..., o_idx] = x[:, -1].index_select(-1, i_idx).to(y.dtype)
    y = y + res
    for bounding in model.boundings:  # in-place clamps on the cloned output: plain differentiable torch ops
        y = bounding(y)
    return y


def model_forward(model, x: Tensor) -> Tensor:
    """``AnemoiModelEncProcDec.forward`` (reference models/encoder_processor_decoder.py:168-233) through the sub-modules'
    own ``forward`` methods -- whatever families the config names (GraphTransformer / GNN mappers and processor)."""
    b, _, ens, g, _ = x.shape
    if ens != 1 and b != 1:
        raise NotImplementedError("an ensemble dimension > 1 only with batch size 1 (the reference repeats the node "
                                  "attributes per batch element only)")
    rows = b * ens
    data, hidden = model._graph_name_data, model._graph_name_hidden
    dtype = runtime.compute_dtype(x)  # under torch.autocast: the autocast dtype
    with torch.autocast(device_type=x.device.type, enabled=False):  # this route picks its precisions itself: the
        # activations are cast once here and every sub-module follows the dtype of what it is handed
        from .layers.mapper import GraphTransformerBaseMapper as _GTMapper

        na = model.node_attributes
        width = x.shape[1] * x.shape[4] + na.attr_ndims[data]
        # bf16 GraphTransformer mappers that fold their embedding: the input is written ONCE, in the layout of their first GEMMs
        augmented = (x.is_cuda and not x.requires_grad and x.dtype == torch.float32 and dtype == torch.bfloat16
                     and isinstance(model.encoder, _GTMapper) and isinstance(model.decoder, _GTMapper)
                     and runtime.embed_fold_enabled(dtype) and hasattr(model.encoder, "emb_nodes_src")
                     and model.encoder.emb_nodes_src.in_features == width
                     and os.environ.get("ANEMOI_AMD_TRAIN_ASSEMBLE", "1") != "0")
        if augmented:
            x_data = _AssembleNodes.apply(x, na.latlons(data), na.trainable_tensors[data].trainable, dtype,
                                          ops.round_up(width + 1, ops.k_multiple(dtype)))
        else:
            x_data = torch.cat([x.permute(0, 2, 3, 1, 4).reshape(rows * g, -1), _node_rows(model, data, rows)], dim=1).to(dtype)
        x_hidden = _node_rows(model, hidden, rows)
        shapes = None  # (single device: the modules ignore shard shapes on this route)
        # As in inference, the mesh rows live in the internal Morton order between encoder and decoder (gather locality of
        # the edge kernels, forward AND backward): only the small attribute table is permuted, the plans relabel the
        # mesh side of every edge.  Graph processors only -- a sliding attention window is defined on the external order.
        from .layers.mapper import GNNBaseMapper, GraphTransformerBaseMapper
        from .layers.processor import GNNProcessor, GraphTransformerProcessor

        inv = None
        graph_proc = isinstance(model.processor, (GraphTransformerProcessor, GNNProcessor))
        if graph_proc and all(isinstance(m, (GraphTransformerBaseMapper, GNNBaseMapper)) for m in (model.encoder, model.decoder)):
            order, inv = model._mesh_order(x.device)
            n_mesh = order.numel()
            idx = order if rows == 1 else (torch.arange(rows, device=order.device)[:, None] * n_mesh + order[None, :]).reshape(-1)
            x_hidden = autograd.permute_rows(x_hidden, idx) if x_hidden.requires_grad else x_hidden.index_select(0, idx)
        x_hidden = x_hidden.to(dtype)

        def run_mapper(mapper, a, c, src_map, dst_map):
            if inv is None and augmented:
                # (a processor that keeps the external mesh order -- the Transformer -- between GraphTransformer mappers: the
                #  grid rows are ``[features | 1 | 0-pad]`` here too, which the mappers' module-level call cannot be told)
                with mapper._offloaded():
                    y = gt_mapper(mapper, a, c, rows, None, None, augmented=True)
                return (a, y) if hasattr(mapper, "emb_nodes_src") else y
            if inv is None:
                return mapper((a, c), rows, shapes)
            if isinstance(mapper, GraphTransformerBaseMapper):
                y = gt_mapper(mapper, a, c, rows, src_map, dst_map, augmented=augmented)
                return (a, y) if hasattr(mapper, "emb_nodes_src") else y  # forward mapper: raw source handed on
            hs, hd = gnn_mapper(mapper, a, c, rows, src_map, dst_map)
            return (hs, hd) if hasattr(mapper, "emb_nodes_src") else hd

        x_data_latent, x_latent = _checkpoint(lambda a, c: run_mapper(model.encoder, a, c, None, inv), x_data, x_hidden)
        if inv is None:
            x_proc = model.processor(x_latent, rows, shapes)
        elif isinstance(model.processor, GraphTransformerProcessor):
            x_proc = gt_processor(model.processor, x_latent, rows, inv)
        else:
            x_proc = gnn_processor(model.processor, x_latent, rows, inv)
        x_latent_proc = x_proc + x_latent
        out = _checkpoint(lambda a, c: run_mapper(model.decoder, a, c, inv, None), x_latent_proc, x_data_latent, last=True)
        return _finish(model, out, x, b, ens, g)


def hierarchical_forward(model, x: Tensor) -> Tensor:
    """``AnemoiModelEncProcDecHierarchical.forward`` (reference models/hierarchical.py:178-308): encoder, level
    processors and downscale mappers to the coarsest level, upscale mappers with skip connections back, decoder."""
    b, _, ens, g, _ = x.shape
    if ens != 1 and b != 1:
        raise NotImplementedError("an ensemble dimension > 1 only with batch size 1")
    rows = b * ens
    data, names = model._graph_name_data, model._graph_hidden_names
    dtype = runtime.compute_dtype(x)

    def first(out):  # GraphTransformer backward mappers return the destination nodes; forward / GNN mappers (src, dst)
        return out[1] if isinstance(out, tuple) else out

    def run(mapper, a, c, last=False):
        return _checkpoint(lambda p, q: mapper((p, q), rows, None), a, c, last=last)

    with torch.autocast(device_type=x.device.type, enabled=False):
        x_data = torch.cat([x.permute(0, 2, 3, 1, 4).reshape(rows * g, -1), _node_rows(model, data, rows)], dim=1).to(dtype)
        x_hidden = {h: _node_rows(model, h, rows).to(dtype) for h in names}
        curr = first(run(model.encoder, x_data, x_hidden[names[0]]))
        x_skip, x_encoded = {}, {}
        for src, dst in zip(names[:-1], names[1:]):  # ---- down (reference :224-249)
            if model.level_process:
                curr = model.down_level_processor[src](curr, rows, None)
            x_skip[src] = curr
            out = run(model.downscale[src], curr, x_hidden[dst])
            x_encoded[src], curr = out if isinstance(out, tuple) else (curr, out)
        if model.level_process:  # coarsest level (reference :252-258)
            curr = model.down_level_processor[names[-1]](curr, rows, None)
        for dst, src in zip(reversed(names[:-1]), reversed(names[1:])):  # ---- up (reference :261-286)
            curr = first(run(model.upscale[src], curr, x_encoded[dst])) + x_skip[dst]
            if model.level_process:
                curr = model.up_level_processor[dst](curr, rows, None)
        out = first(run(model.decoder, curr, x_data, last=True))
        return _finish(model, out, x, b, ens, g)
