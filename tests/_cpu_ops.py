"""CPU stand-ins for ``anemoi_models_amd.ops`` used ONLY by tests of the host logic.

They are built on plain torch + the oracle and are substituted from the tests (monkeypatch); the package never
imports this file.  Purpose: check, without a GPU, that the launch sequences written in the layer mirrors
(weight packing, concatenated GEMMs, CSR plans, K padding, residual placement) compute the reference function.
"""

from __future__ import annotations

import torch
import torch.nn.functional as F

from oracle.pyg_semantics import scatter_sum
from oracle.pyg_semantics import segment_softmax

_ACT = {"Identity": lambda t: t, "GELU": F.gelu, "SiLU": F.silu, "ReLU": F.relu}


def layer_norm(x, weight, bias, eps=1e-5, out=None, residual=None):
    if residual is not None:
        return layer_norm(x, weight, bias, eps, out) + residual
    return _layer_norm_plain(x, weight, bias, eps, out)


def _layer_norm_plain(x, weight, bias, eps=1e-5, out=None):
    return F.layer_norm(x.float(), (x.shape[1],), weight, bias, eps).to(x.dtype)


def layer_norm_with_stats(x, weight, bias, eps=1e-5):
    return layer_norm(x, weight, bias, eps), row_stats(x, eps)


def row_stats(x, eps=1e-5):
    xf = x.float()
    mean = xf.mean(dim=1)
    rstd = torch.rsqrt(xf.var(dim=1, unbiased=False) + eps)
    return torch.stack([rstd, -mean * rstd], dim=1).contiguous()


def linear(x, w, bias=None, *, act="Identity", residual=None, out=None, out_dtype=None, n_out=None, ln=None,
           stats_eps=None):
    assert x.shape[1] == w.shape[1] and w.shape[1] % (128 // x.element_size()) == 0, "K must be slab padded"
    if ln is None:
        pre = F.linear(x.float(), w.float(), bias)
    else:  # the kernel's arithmetic: rstd (x W'^T) + (-mean rstd) colsum + b'
        stats, colsum = ln
        assert stats.shape == (x.shape[0], 2) and colsum.shape[0] >= w.shape[0]
        pre = F.linear(x.float(), w.float()) * stats[:, :1] + stats[:, 1:] * colsum[None, : w.shape[0]]
        if bias is not None:
            pre = pre + bias
    y = _ACT[act](pre)
    if residual is not None:
        y = y + residual.float()
    y = y.to(out_dtype or x.dtype)
    if out is not None:
        out.copy_(y)
        return out
    return y


def linear_dual(x, w, bias, act):
    pre = linear(x, w, bias)
    return pre, _ACT[act](pre.float()).to(pre.dtype)


def edge_attr_csr(a0, a1, perm, ld_out=None, one_col=-1):
    rows = perm.long() % a0.shape[0]
    parts = [a0[rows].float()] + ([] if a1 is None else [a1[rows].float()])
    out = torch.cat(parts, dim=1)
    ld = (out.shape[1] + 3) // 4 * 4 if ld_out is None else ld_out
    out = F.pad(out, (0, ld - out.shape[1]))
    if one_col >= 0:
        out[:, one_col] = 1.0
    return out


def gt_edge_attention_folded(q, k, v, x_r, u, edge_attr, rowptr, col, num_heads, up, out=None, ld_out=None, lse=None, runs=None, sched=None, tiles=None):
    n_dst, c = q.shape
    d = c // num_heads
    dst = torch.repeat_interleave(torch.arange(n_dst), (rowptr[1:] - rowptr[:-1]).long())
    src = col.long()
    qi = q.float().reshape(n_dst, num_heads, d)[dst]
    kj = k.float().reshape(-1, num_heads, d)[src]
    vj = v.float().reshape(-1, num_heads, d)[src]
    ui = u.float().reshape(n_dst, num_heads, up)[dst]
    a = edge_attr.float().unsqueeze(1)  # [E, 1, up]
    score = ((qi * kj).sum(-1) + (ui * a).sum(-1)) / d**0.5
    alpha = segment_softmax(score, dst, n_dst)
    res = scatter_sum(vj * alpha.unsqueeze(-1), dst, n_dst).reshape(n_dst, c)
    t = scatter_sum(a * alpha.unsqueeze(-1), dst, n_dst).reshape(n_dst, num_heads * up)
    if x_r is not None:
        res = res + x_r.float()
    res = torch.cat([res, t], dim=1)
    ld = res.shape[1] if ld_out is None else ld_out
    return F.pad(res, (0, ld - res.shape[1])).to(q.dtype)


def edge_dropout_keep_mask(seed: int, p: float, n_edges: int, heads: int) -> torch.Tensor:
    """The edge kernels' counter-based keep mask (csrc/common.hpp::edge_dropout_keep) restated with torch integer
    arithmetic: ``[E, H]`` of 0 / 1 over the CSR edge positions."""
    m32 = 0xFFFFFFFF
    e = torch.arange(n_edges, dtype=torch.int64).view(-1, 1)
    h = torch.arange(heads, dtype=torch.int64).view(1, -1)
    x = ((e * 0x9E3779B1) & m32) ^ ((h * 0x85EBCA77) & m32) ^ (seed & m32)
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & m32
    x = x ^ (x >> 15)
    thr = min(int(p * 32768.0 + 0.5), 32768)
    return ((x >> 17) >= thr).to(torch.float64) if p < 1.0 else torch.zeros(n_edges, heads, dtype=torch.float64)


def gt_conv(q, k, v, edges_csr, rowptr, col, num_heads, x_r=None, lse=None, dropout_p=0.0, dropout_seed=0, seed_dev=None):
    n_dst, c = q.shape
    d = c // num_heads
    dst = torch.repeat_interleave(torch.arange(n_dst), (rowptr[1:] - rowptr[:-1]).long())
    src = col.long()
    e = edges_csr.float().reshape(-1, num_heads, d)
    kj = k.float().reshape(-1, num_heads, d)[src] + e
    vj = v.float().reshape(-1, num_heads, d)[src] + e
    score = (q.float().reshape(n_dst, num_heads, d)[dst] * kj).sum(-1) / d**0.5
    alpha = segment_softmax(score, dst, n_dst)
    if dropout_p > 0.0:
        seed = int(dropout_seed) + (0 if seed_dev is None else int(seed_dev.reshape(-1)[0]))
        keep = edge_dropout_keep_mask(seed, dropout_p, col.shape[0], num_heads).float()
        alpha = alpha * keep * (1.0 / (1.0 - dropout_p) if dropout_p < 1.0 else 0.0)
    out = scatter_sum(vj * alpha.unsqueeze(-1), dst, n_dst).reshape(n_dst, c)
    return (out if x_r is None else out + x_r.float()).to(q.dtype)


def gt_edge_attention(q, k, v, x_r, edge_attr, edge_dim, w_edge, b_edge, rowptr, col, num_heads, out=None):
    n_dst, c = q.shape
    d = c // num_heads
    dst = torch.repeat_interleave(torch.arange(n_dst), (rowptr[1:] - rowptr[:-1]).long())
    src = col.long()
    e = F.linear(edge_attr[:, :edge_dim], w_edge, b_edge).view(-1, num_heads, d)
    qi = q.float().view(n_dst, num_heads, d)[dst]
    kj = k.float().reshape(-1, num_heads, d)[src] + e
    vj = v.float().reshape(-1, num_heads, d)[src] + e
    alpha = segment_softmax((qi * kj).sum(-1) / d**0.5, dst, n_dst)
    res = scatter_sum(vj * alpha.unsqueeze(-1), dst, n_dst).reshape(n_dst, c)
    if x_r is not None:
        res = res + x_r.float()
    return res.to(q.dtype)


def gather_add_act(t, p_dst, p_src, dst, src, act="Identity", out=None):
    y = _ACT[act](t.float() + p_dst.float()[dst.long()] + p_src.float()[src.long()]).to(t.dtype)
    if out is not None:
        out.copy_(y)
        return out
    return y


def segment_sum(v, rowptr, out=None, cat_with=None):
    n = rowptr.shape[0] - 1
    dst = torch.repeat_interleave(torch.arange(n), (rowptr[1:] - rowptr[:-1]).long())
    y = scatter_sum(v.float(), dst, n).to(v.dtype)
    if cat_with is not None:
        return torch.cat([cat_with, y], dim=1)
    if out is not None:
        out.copy_(y)
        return out
    return y


def mhsa(qkv, batch_size, num_heads, window=-1, out=None, return_lse=False, dropout_p=0.0, dropout_seed=0, head_offset=0,
         heads_total=None, seed_dev=None):
    assert dropout_p == 0.0, "the CPU stand-in has no attention dropout"
    rows, c3 = qkv.shape
    c = c3 // 3
    s_len, d = rows // batch_size, c // num_heads
    q, k, v = (t.float().reshape(batch_size, s_len, num_heads, d).permute(0, 2, 1, 3) for t in qkv.split(c, dim=1))
    sc = q @ k.transpose(-1, -2) / d**0.5
    if window >= 0:
        i = torch.arange(s_len)
        sc = sc.masked_fill((i[:, None] - i[None, :]).abs() > window, float("-inf"))
    o = (torch.softmax(sc, -1) @ v).permute(0, 2, 1, 3).reshape(rows, c).to(qkv.dtype)
    return (o, torch.logsumexp(sc, -1)) if return_lse else o


def assemble_nodes(x, latlons, trainable, batch_size, dtype, ld_out=None, ensemble=1, in_affine=None, rows=None):
    if rows is not None:  # (batch 1, ensemble 1: the node rows of the full matrix)
        return assemble_nodes(x, latlons, trainable, batch_size, dtype, ld_out, ensemble, in_affine).index_select(0, rows)
    parts = []
    if x is not None and in_affine is not None:
        x = x * in_affine[0] + in_affine[1]
    if x is not None:
        b, t, ens, g, v = x.shape
        parts.append(x.permute(0, 2, 3, 1, 4).reshape(b * ens * g, t * v))
        rep = b
    else:
        rep = batch_size
    parts.append(latlons.repeat(rep, 1))
    if trainable is not None:
        parts.append(trainable.detach().repeat(rep, 1))
    out = torch.cat(parts, dim=1)
    ld = out.shape[1] if ld_out is None else ld_out
    return F.pad(out, (0, ld - out.shape[1])).to(dtype)


def prognostic_residual(y, x, out_idx, in_idx):
    y[..., out_idx.long()] += x[:, -1, :, :, in_idx.long()]
    return y


def act_forward(pre, act, residual=None):
    y = {"GELU": F.gelu, "SiLU": F.silu, "ReLU": torch.relu, "Identity": lambda t: t}[act](pre.float()).to(pre.dtype)
    return y if residual is None else y + residual


def finalize_output(y, x, src, in_affine=None, out_affine=None, rows=None):
    last = x[:, -1]
    if rows is not None:  # y holds the rows of these grid nodes only (batch 1, ensemble 1)
        last = last.index_select(2, rows).reshape(y.shape[:-1] + (x.shape[-1],))
    if in_affine is not None:
        last = last * in_affine[0] + in_affine[1]
    cols = torch.nonzero(src >= 0).flatten()
    y[..., cols] += last[..., src[cols].long()]
    if out_affine is not None:
        y.copy_((y - out_affine[1]) / out_affine[0])
    return y


def bound_output(y, op_col, op_lo, op_hi, op_mul, fin=None):
    """Sequential restatement of anemoi_bound_output (host-logic tests only)."""
    for c, lo, hi, m in zip(op_col.tolist(), op_lo.tolist(), op_hi.tolist(), op_mul.tolist()):
        v = y[..., c]
        v = torch.where(v < lo, torch.full_like(v, lo), torch.where(v > hi, torch.full_like(v, hi), v))
        if m >= 0:
            v = v * y[..., m]
        y[..., c] = v
    if fin is not None:
        for c, mul, add in zip(fin[0].tolist(), fin[1].tolist(), fin[2].tolist()):
            y[..., c] = (y[..., c] - add) / mul
    return y


def advance_input(x, y, colmap, forcing=None):
    new = x.roll(-1, dims=1)
    new[:, -1] = x[:, -1]
    for v, m in enumerate(colmap.tolist()):
        if m >= 0:
            new[:, -1, :, :, v] = y[..., m]
        elif m <= -2 and forcing is not None:
            new[:, -1, :, :, v] = forcing[..., -2 - m]
    x.copy_(new)
    return x


def convert_pad(src, dtype, ld_out=None):
    ld = src.shape[1] if ld_out is None else ld_out
    return F.pad(src, (0, ld - src.shape[1])).to(dtype)


def add(a, b, out=None):
    return a + b


def install(monkeypatch):
    import anemoi_models_amd.ops as ops

    for name in ("layer_norm", "layer_norm_with_stats", "row_stats", "linear", "linear_dual", "edge_attr_csr", "gt_edge_attention", "gt_edge_attention_folded",
                 "gt_conv", "gather_add_act", "segment_sum", "mhsa", "assemble_nodes",
                 "prognostic_residual", "finalize_output", "bound_output", "advance_input", "convert_pad", "add", "act_forward"):
        monkeypatch.setattr(ops, name, globals()[name])
