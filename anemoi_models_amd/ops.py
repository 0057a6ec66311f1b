"""Tensor-level wrappers over the C ABI (``include/anemoi_amd.h``).

Every function takes torch tensors that live on an MI355X, passes raw device pointers, sizes and the
current HIP stream to ``libanemoi_amd.so`` and returns torch tensors.  PyTorch is used for device
memory and streams only.  There is deliberately no CPU implementation behind these names: calling
them with CPU tensors raises.  (CPU tests of the host logic substitute this module's functions from
``tests/``; the package itself never does.)
"""

from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

from . import _lib

_DT = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16}

# Optional per-launch timing used by bench.py's roofline leg (never active inside a timed region):
# when PROFILE is a list, every wrapper appends (kernel name, start event, end event, work dict) with the
# events recorded on the stream the kernel is launched on.
PROFILE: Optional[list] = None


class _Timed:
    __slots__ = ("name", "work", "start")

    def __init__(self, name: str, **work) -> None:
        self.name, self.work, self.start = name, work, None

    def __enter__(self):
        if PROFILE is not None:
            self.start = torch.cuda.Event(enable_timing=True)
            self.start.record(torch.cuda.current_stream())
        return self

    def __exit__(self, *exc):
        if self.start is not None:
            end = torch.cuda.Event(enable_timing=True)
            end.record(torch.cuda.current_stream())
            PROFILE.append((self.name, self.start, end, self.work))
        return False


def dtype_code(dtype: torch.dtype) -> int:
    try:
        return _DT[dtype]
    except KeyError:
        raise NotImplementedError(f"compute dtype {dtype} is not supported (float32 or bfloat16)") from None


def k_multiple(dtype: torch.dtype) -> int:
    """The K dimension of a Linear must be a multiple of this many elements (128-byte K-slabs)."""
    return 128 // torch.empty((), dtype=dtype).element_size()


def round_up(n: int, m: int) -> int:
    return (n + m - 1) // m * m


def _dev(*tensors: Optional[Tensor]) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "anemoi_models_amd kernels run on an MI355X only: got a CPU tensor (there is no CPU fallback)"
            )


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> int:
    """hipStream_t of torch's current stream on the current device (the raw getter skips the Stream object)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _rows(t: Tensor) -> Tensor:
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise ValueError(f"expected a row-major 2-D tensor (unit inner stride), got shape {tuple(t.shape)} "
                         f"strides {t.stride()}")
    return t


def _ld(t: Tensor) -> int:
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


def _ptr(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def layer_norm(x: Tensor, weight: Tensor, bias: Tensor, eps: float = 1e-5, out: Optional[Tensor] = None,
               residual: Optional[Tensor] = None) -> Tensor:
    """``LayerNorm(x)`` (``+ residual`` in the same pass: the trailing LayerNorm of an MLP followed by its skip connection)."""
    _dev(x, weight, bias, out, residual)
    _rows(x)
    if out is None:
        out = torch.empty((x.shape[0], x.shape[1]), dtype=x.dtype, device=x.device)
    if residual is not None:
        if residual.dtype != x.dtype or residual.shape != x.shape:
            raise ValueError("layer_norm: the residual must have the shape and dtype of x")
        with _Timed("layer_norm", bytes=3 * x.shape[0] * x.shape[1] * x.element_size()):
            st = _lib.load().anemoi_layer_norm_residual(dtype_code(x.dtype), x.data_ptr(), _ld(x), weight.data_ptr(),
                                                        bias.data_ptr(), residual.data_ptr(), _ld(_rows(residual)),
                                                        out.data_ptr(), _ld(_rows(out)), x.shape[0], x.shape[1], eps, _stream())
        _lib.check(st, "anemoi_layer_norm_residual")
        return out
    with _Timed("layer_norm", bytes=2 * x.shape[0] * x.shape[1] * x.element_size()):
        st = _lib.load().anemoi_layer_norm(dtype_code(x.dtype), x.data_ptr(), _ld(x), weight.data_ptr(),
                                           bias.data_ptr(), out.data_ptr(), _ld(_rows(out)), x.shape[0], x.shape[1],
                                           eps, _stream())
    _lib.check(st, "anemoi_layer_norm")
    return out


def layer_norm_with_stats(x: Tensor, weight: Tensor, bias: Tensor, eps: float = 1e-5):
    """``(LayerNorm(x), row statistics [M, 2] f32 = (rstd, -mean * rstd))`` from ONE pass over ``x`` (the training forward
    keeps the statistics for the backward; ``row_stats`` + ``layer_norm`` read ``x`` twice)."""
    _dev(x, weight, bias)
    _rows(x)
    out = torch.empty((x.shape[0], x.shape[1]), dtype=x.dtype, device=x.device)
    stats = torch.empty((x.shape[0], 2), dtype=torch.float32, device=x.device)
    st = _lib.load().anemoi_layer_norm_stats(dtype_code(x.dtype), x.data_ptr(), _ld(x), weight.data_ptr(), bias.data_ptr(),
                                             out.data_ptr(), _ld(out), stats.data_ptr(), x.shape[0], x.shape[1], eps,
                                             _stream())
    _lib.check(st, "anemoi_layer_norm_stats")
    return out, stats


def _carry_stats(y: Tensor, eps: float, stats: Tensor) -> None:
    """Attach the LayerNorm statistics the producing GEMM's epilogue computed to ``y``.  The record pins the storage
    pointer and torch's in-place version counter of ``y`` at this moment: any later in-place write (or a swapped
    ``.data``) makes :func:`_carried_stats` ignore it, so stale statistics cannot be consumed silently."""
    y._anemoi_row_stats = (eps, stats, y.data_ptr(), y._version)


def _carried_stats(x: Tensor, eps: float) -> Optional[Tensor]:
    rec = getattr(x, "_anemoi_row_stats", None)
    if rec is None:
        return None
    r_eps, stats, ptr, version = rec
    if r_eps != eps or stats.shape[0] != x.shape[0] or ptr != x.data_ptr() or version != x._version:
        return None
    return stats


def row_stats(x: Tensor, eps: float = 1e-5) -> Tensor:
    """LayerNorm statistics of the rows of ``x``: ``[M, 2]`` f32 = ``(rstd, -mean * rstd)`` (see ``linear(ln=...)``)."""
    _dev(x)
    _rows(x)
    carried = _carried_stats(x, eps)  # left by linear(..., stats_eps=eps), which produced x
    if carried is not None:
        return carried
    out = torch.empty((x.shape[0], 2), dtype=torch.float32, device=x.device)
    with _Timed("row_stats", bytes=x.shape[0] * x.shape[1] * x.element_size()):
        st = _lib.load().anemoi_row_stats(dtype_code(x.dtype), x.data_ptr(), _ld(x), out.data_ptr(), x.shape[0],
                                          x.shape[1], eps, _stream())
    _lib.check(st, "anemoi_row_stats")
    return out


def linear(x: Tensor, w: Tensor, bias: Optional[Tensor] = None, *, act: str = "Identity",
           residual: Optional[Tensor] = None, out: Optional[Tensor] = None, out_dtype: Optional[torch.dtype] = None,
           n_out: Optional[int] = None, ln=None, stats_eps: Optional[float] = None) -> Tensor:
    """``act(x @ w.T + bias) + residual``; ``w`` is ``[N, K]`` in x's dtype with K already padded like x.

    ``stats_eps``: the result is about to enter a LayerNorm with this epsilon -- its row statistics are produced by the
    GEMM's epilogue (``anemoi_linear_stats``) and travel with the returned tensor: ``row_stats(result, stats_eps)`` then
    costs nothing.  Identity activation, same dtype in and out; silently ignored otherwise.

    ``ln=(stats, colsum)`` folds the LayerNorm of ``x`` into the product: ``x`` is the un-normalised input,
    ``stats = row_stats(x)``, ``w`` / ``bias`` / ``colsum`` come from ``runtime.fold_layer_norm``.
    """
    _dev(x, w, bias, residual, out)
    _rows(x)
    if w.dtype != x.dtype or not w.is_contiguous():
        raise ValueError("linear: weight must be contiguous and in the activation dtype")
    n = w.shape[0] if n_out is None else n_out
    k = w.shape[1]
    if x.shape[1] != k:
        raise ValueError(f"linear: x has {x.shape[1]} columns, weight expects {k}")
    if out is None:
        out = torch.empty((x.shape[0], n), dtype=out_dtype or x.dtype, device=x.device)
    if act not in _lib.ACT_CODES:
        raise RuntimeError(f"activation {act} is not supported by the fused Linear kernel")
    if x.shape[0] == 0:  # (an edge set without edges, an empty shard: torch hands out null pointers for empty tensors)
        return out
    alg = (x.shape[0] * k + n * k) * x.element_size() + x.shape[0] * n * (out.element_size() + (
        0 if residual is None else residual.element_size()))
    fuse_stats = stats_eps is not None and act == "Identity" and out.dtype == x.dtype and _rows(out).shape[1] == n
    with _Timed("linear", flops=2 * x.shape[0] * n * k, bytes=alg, m=x.shape[0], n=n, k=k):
        if fuse_stats:
            stats_in = colsum = None
            if ln is not None:
                stats_in, colsum = ln
                _dev(stats_in, colsum)
            m_rows = x.shape[0]
            ws = torch.empty((m_rows * max(n // 128, 1), 2), dtype=torch.float32, device=x.device)
            stats_out = torch.empty((m_rows, 2), dtype=torch.float32, device=x.device)
            st = _lib.load().anemoi_linear_stats(
                dtype_code(x.dtype), x.data_ptr(), _ld(x), w.data_ptr(), _ptr(bias), _ptr(colsum), _ptr(stats_in),
                _ptr(residual), 0 if residual is None else _ld(_rows(residual)), out.data_ptr(), _ld(_rows(out)), m_rows,
                n, k, ws.data_ptr(), ws.numel() * 4, stats_eps, stats_out.data_ptr(), _stream())
            _carry_stats(out, stats_eps, stats_out)
        elif ln is None:
            st = _lib.load().anemoi_linear(
                dtype_code(x.dtype), dtype_code(out.dtype), x.data_ptr(), _ld(x), w.data_ptr(), _ptr(bias),
                _ptr(residual), 0 if residual is None else _ld(_rows(residual)), out.data_ptr(), _ld(_rows(out)),
                x.shape[0], n, k, _lib.ACT_CODES[act], _stream())
        else:
            stats, colsum = ln
            _dev(stats, colsum)
            if stats.shape != (x.shape[0], 2) or stats.dtype != torch.float32 or not stats.is_contiguous():
                raise ValueError("linear: ln stats must be the contiguous [M, 2] f32 result of row_stats(x)")
            if colsum.numel() < n or colsum.dtype != torch.float32 or not colsum.is_contiguous():
                raise ValueError("linear: ln colsum must be a contiguous f32 vector with one entry per output column")
            st = _lib.load().anemoi_linear_ln(
                dtype_code(x.dtype), dtype_code(out.dtype), x.data_ptr(), _ld(x), w.data_ptr(), _ptr(bias),
                colsum.data_ptr(), stats.data_ptr(), _ptr(residual),
                0 if residual is None else _ld(_rows(residual)), out.data_ptr(), _ld(_rows(out)), x.shape[0], n, k,
                _lib.ACT_CODES[act], _stream())
    _lib.check(st, "anemoi_linear" if ln is None else "anemoi_linear_ln")
    return out


def edge_attr_csr(a0: Tensor, a1: Optional[Tensor], perm: Tensor, ld_out: Optional[int] = None,
                  one_col: int = -1) -> Tensor:
    """Edge attributes ``[a0 | a1]`` gathered into CSR order (rows ``perm[e] % a0.shape[0]``), f32, zero padded.

    ``one_col >= 0`` writes a constant 1 into that (padding) column: the bias carrier of the folded edge kernel.
    """
    _dev(a0, a1, perm)
    d0 = a0.shape[1]
    d1 = 0 if a1 is None else a1.shape[1]
    ld = round_up(d0 + d1, 4) if ld_out is None else ld_out
    a0 = a0.contiguous().float()
    a1 = None if a1 is None else a1.contiguous().float()
    out = torch.empty((perm.shape[0], ld), dtype=torch.float32, device=a0.device)
    if perm.shape[0] == 0:
        return out
    st = _lib.load().anemoi_edge_attr_csr(a0.data_ptr(), d0, _ptr(a1), d1, a0.shape[0], perm.data_ptr(),
                                          out.data_ptr(), ld, one_col, perm.shape[0], _stream())
    _lib.check(st, "anemoi_edge_attr_csr")
    return out


def gt_edge_attention(q: Tensor, k: Tensor, v: Tensor, x_r: Optional[Tensor], edge_attr: Tensor, edge_dim: int,
                      w_edge: Tensor, b_edge: Tensor, rowptr: Tensor, col: Tensor, num_heads: int,
                      out: Optional[Tensor] = None) -> Tensor:
    """Fused gather -> lin_edge -> score -> segment softmax -> weighted scatter-sum (+ x_r) over a dst-CSR graph."""
    _dev(q, k, v, x_r, edge_attr, w_edge, b_edge, rowptr, col, out)
    n_dst, c = _rows(q).shape
    if _ld(_rows(k)) != _ld(_rows(v)):
        raise ValueError("gt_edge_attention: k and v must share their leading dimension")
    if out is None:
        out = torch.empty((n_dst, c), dtype=q.dtype, device=q.device)
    if rowptr.dtype != torch.int32 or col.dtype != torch.int32 or rowptr.shape[0] != n_dst + 1:
        raise ValueError("gt_edge_attention: rowptr/col must be int32 with rowptr of length n_dst + 1")
    if col.shape[0] == 0:  # graph without edges: the C ABI still wants valid (never dereferenced) pointers
        col = torch.zeros(1, dtype=torch.int32, device=q.device)
        edge_attr = torch.zeros((1, max(4, round_up(edge_dim, 4))), dtype=torch.float32, device=q.device)
    # algorithmic bytes per SURVEY.md section 8d: q + out over the destinations, k + v over the sources (each once),
    # 48 B of raw attributes + 4 B column index per edge, the row pointers
    alg_bytes = (2 * n_dst + 2 * k.shape[0]) * c * q.element_size() + col.shape[0] * 52 + (n_dst + 1) * 4
    with _Timed("gt_edge_attention", bytes=alg_bytes, n_dst=n_dst, n_src=k.shape[0], edges=col.shape[0]):
        st = _lib.load().anemoi_gt_edge_attention(
            dtype_code(q.dtype), q.data_ptr(), _ld(q), k.data_ptr(), v.data_ptr(), _ld(k), _ptr(x_r),
            0 if x_r is None else _ld(_rows(x_r)), edge_attr.data_ptr(),
            edge_attr.stride(0) if edge_attr.shape[0] > 1 else edge_attr.shape[1], edge_dim, w_edge.data_ptr(),
            b_edge.data_ptr(), rowptr.data_ptr(), col.data_ptr(), out.data_ptr(), _ld(_rows(out)), n_dst, c,
            num_heads, _stream())
    _lib.check(st, "anemoi_gt_edge_attention")
    return out


def gt_edge_attention_folded(q: Tensor, k: Tensor, v: Tensor, x_r: Optional[Tensor], u: Tensor, edge_attr: Tensor,
                             rowptr: Tensor, col: Tensor, num_heads: int, up: int, out: Optional[Tensor] = None,
                             ld_out: Optional[int] = None, lse: Optional[Tensor] = None, runs=None, sched=None,
                             tiles=None) -> Tensor:
    """Edge phase with lin_edge folded away: returns ``[n_dst, ld_out]`` = ``[sum alpha v (+ x_r) | t (H*up) | 0-pad]``.

    ``u`` is ``[n_dst, H*up]`` (extra columns of the q/k/v GEMM), ``edge_attr`` ``[E, up]`` f32 in CSR order with the
    constant-1 column.  Columns beyond ``C + H*up`` (K padding for the projection GEMM) are zero filled.  ``lse``
    (optional f32 ``[n_dst, H]``, contiguous) receives the softmax normaliser per destination and head (training).
    ``runs`` (``EdgePlan.runs3()``, uniform-degree-3 graphs): destinations that share their three sources share one gather
    of them -- ``(grp_ptr, grp_perm, grp_dst)``: all destinations of a source triple (``anemoi_gt_edge_attention_folded_groups``),
    ``(run_ptr, perm)``: the consecutive ones (``anemoi_gt_edge_attention_folded_runs``).  ``sched`` (``EdgePlan.schedule()``: int32
    ``[8, slots, steps]``): the destination schedule of ``anemoi_gt_edge_attention_folded_sched`` -- balanced wave slots, index
    chain resolved one destination ahead; bit-identical to the plain kernel.  ``tiles`` (``EdgePlan.tiles()``: a
    ``runtime.EdgeTiles``): the LDS-tile kernel ``anemoi_gt_edge_attention_folded_tiles`` -- every source row staged once per
    tile of <= 32 destinations; bit-identical to the plain kernel.  ``runs`` wins over ``tiles`` wins over ``sched``.
    """
    _dev(q, k, v, x_r, u, edge_attr, rowptr, col, out, lse)
    if lse is not None and (lse.dtype != torch.float32 or not lse.is_contiguous()
                            or tuple(lse.shape) != (_rows(q).shape[0], num_heads)):
        raise ValueError("gt_edge_attention_folded: lse must be a contiguous f32 [n_dst, H] tensor")
    n_dst, c = _rows(q).shape
    if _ld(_rows(k)) != _ld(_rows(v)):
        raise ValueError("gt_edge_attention_folded: k and v must share their leading dimension")
    width = c + num_heads * up
    ld = width if ld_out is None else ld_out
    if out is None:
        out = torch.empty((n_dst, ld), dtype=q.dtype, device=q.device)
        if ld > width:
            out[:, width:].zero_()
    if rowptr.dtype != torch.int32 or col.dtype != torch.int32 or rowptr.shape[0] != n_dst + 1:
        raise ValueError("gt_edge_attention_folded: rowptr/col must be int32 with rowptr of length n_dst + 1")
    if edge_attr.shape[0] != col.shape[0] or (col.shape[0] > 0 and (edge_attr.shape[1] != up or
                                                                      not edge_attr.is_contiguous())):
        raise ValueError("gt_edge_attention_folded: edge_attr must be contiguous [E, up]")
    if col.shape[0] == 0:
        col = torch.zeros(1, dtype=torch.int32, device=q.device)
        edge_attr = torch.zeros((1, up), dtype=torch.float32, device=q.device)
    alg_bytes = (2 * n_dst + 2 * k.shape[0]) * c * q.element_size() + col.shape[0] * 52 + (n_dst + 1) * 4
    # (+ the operands this kernel moves because of its fusions -- x_r read, u read, t written -- which SURVEY 8d's figure does
    #  not count: reported beside the roofline fraction, never instead of it)
    fused_bytes = alg_bytes + n_dst * ((0 if x_r is None else c) + 2 * num_heads * up) * q.element_size()
    with _Timed("gt_edge_attention", bytes=alg_bytes, fused_bytes=fused_bytes, n_dst=n_dst, n_src=k.shape[0], edges=col.shape[0]):
        if runs is not None and len(runs) == 3:  # groups of destinations that share their three sources
            grp_ptr, grp_perm, grp_dst = runs
            _dev(grp_ptr, grp_perm, grp_dst)
            if (grp_ptr.dtype != torch.int32 or grp_perm.dtype != torch.int32 or grp_dst.dtype != torch.int32
                    or grp_perm.shape[0] != n_dst or grp_dst.shape[0] != n_dst):
                raise ValueError("gt_edge_attention_folded: runs = (int32 grp_ptr [n_groups + 1], int32 grp_perm [n_dst], "
                                 "int32 grp_dst [n_dst])")
            st = _lib.load().anemoi_gt_edge_attention_folded_groups(
                dtype_code(q.dtype), q.data_ptr(), _ld(q), k.data_ptr(), v.data_ptr(), _ld(_rows(k)), _ptr(x_r),
                0 if x_r is None else _ld(_rows(x_r)), u.data_ptr(), _ld(_rows(u)), edge_attr.data_ptr(), up,
                rowptr.data_ptr(), col.data_ptr(), grp_ptr.data_ptr(), grp_dst.data_ptr(), grp_perm.data_ptr(),
                grp_ptr.shape[0] - 1, _rows(k).shape[0], out.data_ptr(), _ld(_rows(out)), _ptr(lse), n_dst, c, num_heads,
                _stream())
        elif runs is not None:
            run_ptr, perm = runs
            _dev(run_ptr, perm)
            if run_ptr.dtype != torch.int32 or perm.dtype != torch.int32 or perm.shape[0] != run_ptr.shape[0] - 1:
                raise ValueError("gt_edge_attention_folded: runs = (int32 run_ptr [n_runs + 1], int32 perm [n_runs])")
            st = _lib.load().anemoi_gt_edge_attention_folded_runs(
                dtype_code(q.dtype), q.data_ptr(), _ld(q), k.data_ptr(), v.data_ptr(), _ld(_rows(k)), _ptr(x_r),
                0 if x_r is None else _ld(_rows(x_r)), u.data_ptr(), _ld(_rows(u)), edge_attr.data_ptr(), up,
                rowptr.data_ptr(), col.data_ptr(), run_ptr.data_ptr(), perm.data_ptr(), run_ptr.shape[0] - 1,
                out.data_ptr(), _ld(_rows(out)), _ptr(lse), n_dst, c, num_heads, _stream())
        elif tiles is not None:
            _dev(tiles.hdr, tiles.dst, tiles.src, tiles.slot, tiles.xcd)
            st = _lib.load().anemoi_gt_edge_attention_folded_tiles(
                dtype_code(q.dtype), q.data_ptr(), _ld(q), k.data_ptr(), v.data_ptr(), _ld(_rows(k)), _ptr(x_r),
                0 if x_r is None else _ld(_rows(x_r)), u.data_ptr(), _ld(_rows(u)), edge_attr.data_ptr(), up,
                rowptr.data_ptr(), col.data_ptr(), tiles.hdr.data_ptr(), tiles.dst.data_ptr(), tiles.src.data_ptr(),
                tiles.slot.data_ptr(), tiles.xcd.data_ptr(), tiles.max_tiles_per_xcd, tiles.src_cap, tiles.edge_cap,
                _rows(k).shape[0], col.shape[0], out.data_ptr(), _ld(_rows(out)), _ptr(lse), n_dst, c, num_heads, _stream())
        elif sched is not None:  # (the entry point itself falls back to the plain kernel beyond 32-bit row offsets)
            _dev(sched)
            if sched.dtype != torch.int32 or sched.dim() != 3 or sched.shape[0] != 8 or not sched.is_contiguous():
                raise ValueError("gt_edge_attention_folded: sched = contiguous int32 [8, slots, steps]")
            st = _lib.load().anemoi_gt_edge_attention_folded_sched(
                dtype_code(q.dtype), q.data_ptr(), _ld(q), k.data_ptr(), v.data_ptr(), _ld(_rows(k)), _ptr(x_r),
                0 if x_r is None else _ld(_rows(x_r)), u.data_ptr(), _ld(_rows(u)), edge_attr.data_ptr(), up,
                rowptr.data_ptr(), col.data_ptr(), sched.data_ptr(), sched.shape[1], sched.shape[2], _rows(k).shape[0],
                col.shape[0], out.data_ptr(), _ld(_rows(out)), _ptr(lse), n_dst, c, num_heads, _stream())
        else:
            st = _lib.load().anemoi_gt_edge_attention_folded(
                dtype_code(q.dtype), q.data_ptr(), _ld(q), k.data_ptr(), v.data_ptr(), _ld(_rows(k)), _ptr(x_r),
                0 if x_r is None else _ld(_rows(x_r)), u.data_ptr(), _ld(_rows(u)), edge_attr.data_ptr(), up,
                rowptr.data_ptr(), col.data_ptr(), out.data_ptr(), _ld(_rows(out)), _ptr(lse), n_dst, c, num_heads, _stream())
    _lib.check(st, "anemoi_gt_edge_attention_folded")
    return out


def gt_conv(q: Tensor, k: Tensor, v: Tensor, edges_csr: Tensor, rowptr: Tensor, col: Tensor, num_heads: int,
            x_r: Optional[Tensor] = None, lse: Optional[Tensor] = None, dropout_p: float = 0.0, dropout_seed: int = 0,
            seed_dev: Optional[Tensor] = None) -> Tensor:
    """``GraphTransformerConv`` with explicit per-edge features (reference layers/conv.py:98-142): ``q [n_dst, C]``,
    ``k, v [n_src, C]``, ``edges_csr [E, C]`` in the CSR order of ``(rowptr, col)``; returns ``[n_dst, C]`` (``+ x_r``).
    ``lse`` (optional f32 ``[n_dst, H]``) receives the softmax normaliser (training).  ``dropout_p`` / ``dropout_seed`` /
    ``seed_dev``: the conv's dropout of the attention weights in training mode (layers/conv.py:140), mask = hash(CSR edge
    position, head, seed) as for :func:`mhsa`."""
    if not 0.0 <= dropout_p <= 1.0:
        raise ValueError(f"dropout probability has to be between 0 and 1, but got {dropout_p}")
    _dev(q, k, v, edges_csr, rowptr, col, x_r, lse)
    n_dst, c = _rows(q).shape
    if _ld(_rows(k)) != _ld(_rows(v)):
        raise ValueError("gt_conv: k and v must share their leading dimension")
    if rowptr.dtype != torch.int32 or col.dtype != torch.int32 or rowptr.shape[0] != n_dst + 1:
        raise ValueError("gt_conv: rowptr/col must be int32 with rowptr of length n_dst + 1")
    if edges_csr.shape[0] != col.shape[0] or edges_csr.dtype != q.dtype:
        raise ValueError("gt_conv: edges must be [E, C] in the activation dtype")
    out = torch.empty((n_dst, c), dtype=q.dtype, device=q.device)
    if col.shape[0] == 0:
        if lse is not None:
            lse.fill_(float("-inf"))
        return out.zero_() if x_r is None else out.copy_(x_r)
    alg_bytes = (2 * n_dst + 2 * k.shape[0] + col.shape[0]) * c * q.element_size() + col.shape[0] * 4 + (n_dst + 1) * 4
    with _Timed("gt_conv", bytes=alg_bytes, n_dst=n_dst, n_src=k.shape[0], edges=col.shape[0]):
        st = _lib.load().anemoi_gt_conv(dtype_code(q.dtype), q.data_ptr(), _ld(q), k.data_ptr(), v.data_ptr(), _ld(_rows(k)),
                                        edges_csr.data_ptr(), _ld(_rows(edges_csr)), _ptr(x_r),
                                        0 if x_r is None else _ld(_rows(x_r)), rowptr.data_ptr(), col.data_ptr(),
                                        out.data_ptr(), _ld(out), _ptr(lse), n_dst, c, num_heads, float(dropout_p),
                                        int(dropout_seed) & 0xFFFFFFFF, _seed_dev_ptr(seed_dev, q), _stream())
    _lib.check(st, "anemoi_gt_conv")
    return out


def gather_add_act(t: Tensor, p_dst: Tensor, p_src: Tensor, dst: Tensor, src: Tensor, act: str = "Identity",
                   out: Optional[Tensor] = None) -> Tensor:
    """``act(t[e] + p_dst[dst[e]] + p_src[src[e]])`` per edge row (first edge-MLP layer of the GNN block)."""
    _dev(t, p_dst, p_src, dst, src, out)
    e, c = _rows(t).shape
    if out is None:
        out = torch.empty((e, c), dtype=t.dtype, device=t.device)
    if e == 0:
        return out
    if dst.dtype != torch.int32 or src.dtype != torch.int32 or dst.shape[0] != e or src.shape[0] != e:
        raise ValueError("gather_add_act: dst / src must be int32 [E]")
    # compulsory bytes: t read + result written per edge, each node table once, the two index lists
    with _Timed("gather_add_act", bytes=(2 * e + p_dst.shape[0] + p_src.shape[0]) * c * t.element_size() + 8 * e):
        st = _lib.load().anemoi_gather_add_act(dtype_code(t.dtype), t.data_ptr(), _ld(t), p_dst.data_ptr(),
                                               _ld(_rows(p_dst)), p_src.data_ptr(), _ld(_rows(p_src)), dst.data_ptr(),
                                               src.data_ptr(), out.data_ptr(), _ld(_rows(out)), e, c,
                                               _lib.ACT_CODES[act], _stream())
    _lib.check(st, "anemoi_gather_add_act")
    return out


def segment_sum(v: Tensor, rowptr: Tensor, out: Optional[Tensor] = None, cat_with: Optional[Tensor] = None) -> Tensor:
    """Sum of the rows of every CSR segment: ``out[i] = v[rowptr[i]:rowptr[i+1]].sum(0)`` (scatter-sum over dst).
    ``cat_with`` = ``x [n_dst, C]``: returns ``[n_dst, 2C] = [x | sums]`` -- the node MLP's input -- from the same pass."""
    _dev(v, rowptr, out, cat_with)
    n_dst = rowptr.shape[0] - 1
    c = _rows(v).shape[1]
    if cat_with is not None:
        x = _rows(cat_with)
        if x.shape != (n_dst, c) or x.dtype != v.dtype or out is not None:
            raise ValueError("segment_sum: cat_with must be [n_dst, C] of v's dtype (and no out=)")
        out = torch.empty((n_dst, 2 * c), dtype=v.dtype, device=v.device)
        if v.shape[0] == 0:
            out[:, :c].copy_(x)
            out[:, c:].zero_()
            return out
        with _Timed("segment_sum", bytes=(v.shape[0] + 3 * n_dst) * c * v.element_size()):
            st = _lib.load().anemoi_segment_sum_cat(dtype_code(v.dtype), v.data_ptr(), _ld(v), rowptr.data_ptr(),
                                                    x.data_ptr(), _ld(x), out.data_ptr(), 2 * c, n_dst, c, _stream())
        _lib.check(st, "anemoi_segment_sum_cat")
        return out
    if out is None:
        out = torch.empty((n_dst, c), dtype=v.dtype, device=v.device)
    if v.shape[0] == 0:
        return out.zero_()
    with _Timed("segment_sum", bytes=(v.shape[0] + n_dst) * c * v.element_size()):
        st = _lib.load().anemoi_segment_sum(dtype_code(v.dtype), v.data_ptr(), _ld(v), rowptr.data_ptr(),
                                            out.data_ptr(), _ld(_rows(out)), n_dst, c, _stream())
    _lib.check(st, "anemoi_segment_sum")
    return out


def _seed_dev_ptr(seed_dev: Optional[Tensor], like: Tensor):
    if seed_dev is None:
        return None
    if not seed_dev.is_cuda or seed_dev.device != like.device or seed_dev.numel() < 1 or seed_dev.element_size() < 4:
        raise ValueError("dropout seed_dev must be an integer tensor (>= 32 bits per element) on the input's device")
    return seed_dev.data_ptr()


def mhsa(qkv: Tensor, batch_size: int, num_heads: int, window: int = -1, out: Optional[Tensor] = None,
         return_lse: bool = False, dropout_p: float = 0.0, dropout_seed: int = 0, head_offset: int = 0,
         heads_total: int = 0, seed_dev: Optional[Tensor] = None):
    """Multi-head self attention on the fused ``lin_qkv`` output ``[B*S, 3C]`` -> ``[B*S, C]`` (heads concatenated).
    ``return_lse``: also the f32 ``[B, H, S]`` log-sum-exp of the scaled scores (the backward's input).
    ``dropout_p`` / ``dropout_seed``: attention dropout (training mode of the reference), mask = hash(index, seed);
    ``head_offset`` / ``heads_total``: this call holds the heads ``head_offset ... + num_heads`` of ``heads_total`` (a head
    shard of a model group draws the mask of the unsharded attention; 0 = all heads).  ``seed_dev``: a device integer
    whose low 32 bits the kernels add to ``dropout_seed`` when they run (``runtime.DeviceDropout``: the part of the seed a
    captured HIP graph advances between replays)."""
    _dev(qkv, out)
    rows, c3 = _rows(qkv).shape
    c = c3 // 3
    s_len = rows // batch_size
    d = c // num_heads
    if out is None:
        out = torch.empty((rows, c), dtype=qkv.dtype, device=qkv.device)
    lib = _lib.load()
    code = dtype_code(qkv.dtype)
    ws_bytes = lib.anemoi_mhsa_workspace_bytes(code, batch_size, s_len, num_heads, d)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=qkv.device) if ws_bytes > 0 else None
    lse = torch.empty((batch_size, num_heads, s_len), dtype=torch.float32, device=qkv.device) if return_lse else None
    with _Timed("mhsa", flops=4 * batch_size * num_heads * s_len * s_len * d, s=s_len, h=num_heads, d=d):
        st = lib.anemoi_mhsa(code, qkv.data_ptr(), _ld(qkv), out.data_ptr(), _ld(_rows(out)), _ptr(ws), _ptr(lse),
                             batch_size, s_len, num_heads, d, window, float(dropout_p), int(dropout_seed) & 0xFFFFFFFF,
                             _seed_dev_ptr(seed_dev, qkv), int(head_offset), int(heads_total), _stream())
    _lib.check(st, "anemoi_mhsa")
    return (out, lse) if return_lse else out


def mhsa_backward(qkv: Tensor, out: Tensor, dout: Tensor, lse: Tensor, batch_size: int, num_heads: int,
                  window: int = -1, dropout_p: float = 0.0, dropout_seed: int = 0, head_offset: int = 0,
                  heads_total: int = 0, use_mfma: bool = True, seed_dev: Optional[Tensor] = None) -> Tensor:
    """``d qkv`` ``[B*S, 3C]`` of :func:`mhsa` from the forward's output and log-sum-exp (``anemoi_mhsa_backward``).
    ``use_mfma=False`` withholds the workspace: the call then takes the VALU kernels (plain HIP, any head size / dtype) --
    the yardstick the MFMA route's hand-scheduled kernels are held against in the tests."""
    _dev(qkv, out, dout, lse)
    rows, c3 = _rows(qkv).shape
    c = c3 // 3
    s_len = rows // batch_size
    dqkv = torch.empty((rows, c3), dtype=qkv.dtype, device=qkv.device)
    delta = torch.empty((batch_size, num_heads, s_len), dtype=torch.float32, device=qkv.device)
    if lse.dtype != torch.float32 or not lse.is_contiguous() or lse.numel() != delta.numel():
        raise ValueError("mhsa_backward: lse must be the contiguous f32 [B, H, S] output of mhsa(return_lse=True)")
    lib = _lib.load()
    ws_bytes = lib.anemoi_mhsa_backward_workspace_bytes(dtype_code(qkv.dtype), batch_size, s_len, num_heads, c // num_heads)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=qkv.device) if ws_bytes > 0 and use_mfma else None
    st = lib.anemoi_mhsa_backward(dtype_code(qkv.dtype), qkv.data_ptr(), _ld(qkv), out.data_ptr(), _ld(_rows(out)),
                                  dout.data_ptr(), _ld(_rows(dout)), lse.data_ptr(), delta.data_ptr(), dqkv.data_ptr(), c3,
                                  _ptr(ws), batch_size, s_len, num_heads, c // num_heads, window, float(dropout_p),
                                  int(dropout_seed) & 0xFFFFFFFF, _seed_dev_ptr(seed_dev, qkv), int(head_offset),
                                  int(heads_total), _stream())
    _lib.check(st, "anemoi_mhsa_backward")
    return dqkv


def assemble_nodes(x: Optional[Tensor], latlons: Tensor, trainable: Optional[Tensor], batch_size: int,
                   dtype: torch.dtype, ld_out: Optional[int] = None, ensemble: int = 1, in_affine=None,
                   rows: Optional[Tensor] = None) -> Tensor:
    """Rows ``(b, ens, g)`` of ``[x (time-major) | latlons | trainable | 0-pad]`` in ``dtype``.  ``in_affine`` =
    ``(mul, add)`` f32 ``[V]``: ``x`` is the raw state, normalised as ``x * mul + add`` while it is read.
    ``rows`` (int64 node ids; batch 1, ensemble 1): only those nodes' rows, in that order (``anemoi_assemble_node_rows``)."""
    _dev(x, latlons, trainable, rows)
    mul = add = None
    if in_affine is not None and x is not None:
        mul, add = (t.contiguous().float() for t in in_affine)
        _dev(mul, add)
        if mul.numel() != x.shape[-1] or add.numel() != x.shape[-1]:
            raise ValueError("assemble_nodes: in_affine needs one (mul, add) pair per input variable")
    g = latlons.shape[0]
    n_ll = latlons.shape[1]
    n_tr = 0 if trainable is None else trainable.shape[1]
    if x is not None:
        b, t, ens, gx, v = x.shape
        if gx != g or b != batch_size:
            raise ValueError(f"assemble_nodes: x has shape {tuple(x.shape)} but the graph has {g} nodes")
        x = x.contiguous().float()
    else:
        b, t, ens, v = batch_size, 0, ensemble, 0
    width = t * v + n_ll + n_tr
    ld = width if ld_out is None else ld_out
    latlons = latlons.contiguous().float()
    trainable = None if trainable is None else trainable.contiguous().float()
    if rows is not None:
        if b != 1 or ens != 1 or rows.dtype != torch.int64 or rows.dim() != 1:
            raise ValueError("assemble_nodes: rows needs batch 1, ensemble 1 and a 1-d int64 id list")
        rows = rows.contiguous()
        out = torch.empty((rows.shape[0], ld), dtype=dtype, device=latlons.device)
        st = _lib.load().anemoi_assemble_node_rows(dtype_code(dtype), _ptr(x), t, g, v, latlons.data_ptr(), n_ll,
                                                   _ptr(trainable), n_tr, rows.data_ptr(), rows.shape[0], out.data_ptr(),
                                                   ld, _ptr(mul), _ptr(add), _stream())
        _lib.check(st, "anemoi_assemble_node_rows")
        return out
    out = torch.empty((b * ens * g, ld), dtype=dtype, device=latlons.device)
    st = _lib.load().anemoi_assemble_nodes(dtype_code(dtype), _ptr(x), b, t, ens, g, v, latlons.data_ptr(), n_ll,
                                           _ptr(trainable), n_tr, out.data_ptr(), ld, _ptr(mul), _ptr(add), _stream())
    _lib.check(st, "anemoi_assemble_nodes")
    return out


def finalize_output(y: Tensor, x: Tensor, src: Tensor, in_affine=None, out_affine=None,
                    rows: Optional[Tensor] = None) -> Tensor:
    """In place on the f32 output ``y`` ``[B, Ens, G, V_out]``: prognostic residual from the last time slice of ``x``
    (``src`` int32 ``[V_out]``: input column per output column, -1 = none; ``in_affine`` normalises a raw ``x`` on the
    fly) and, with ``out_affine = (mul, add)``, the de-normalisation ``(y - add) / mul``.  ``rows`` (int64 node ids; batch
    1, ensemble 1): ``y`` holds only those nodes' rows ``[..., len(rows), V_out]`` (``anemoi_finalize_output_rows``)."""
    _dev(y, x, src, rows)
    if y.dtype != torch.float32 or not y.is_contiguous():
        raise ValueError("finalize_output: y must be contiguous float32")
    x = x.contiguous().float()
    b, t, ens, g, v_in = x.shape
    if src.dtype != torch.int32 or src.numel() != y.shape[-1]:
        raise ValueError("finalize_output: src must be int32 with one entry per output column")
    im = ia = om = oa = None
    if in_affine is not None:
        im, ia = (t_.contiguous().float() for t_ in in_affine)
    if out_affine is not None:
        om, oa = (t_.contiguous().float() for t_ in out_affine)
    _dev(im, ia, om, oa)
    if rows is not None:
        if b != 1 or ens != 1 or rows.dtype != torch.int64 or rows.dim() != 1 or y.numel() != rows.shape[0] * y.shape[-1]:
            raise ValueError("finalize_output: rows needs batch 1, ensemble 1, a 1-d int64 id list and one y row per id")
        rows = rows.contiguous()
        st = _lib.load().anemoi_finalize_output_rows(y.data_ptr(), y.shape[-1], x.data_ptr(), t, g, v_in, src.data_ptr(),
                                                     rows.data_ptr(), rows.shape[0], _ptr(im), _ptr(ia), _ptr(om), _ptr(oa),
                                                     _stream())
        _lib.check(st, "anemoi_finalize_output_rows")
        return y
    st = _lib.load().anemoi_finalize_output(y.data_ptr(), y.shape[-1], x.data_ptr(), b, t, ens, g, v_in, src.data_ptr(),
                                            _ptr(im), _ptr(ia), _ptr(om), _ptr(oa), _stream())
    _lib.check(st, "anemoi_finalize_output")
    return y


def bound_output(y: Tensor, op_col: Tensor, op_lo: Tensor, op_hi: Tensor, op_mul: Tensor,
                 fin: Optional[tuple] = None) -> Tensor:
    """In place on the f32 output ``y`` ``[..., V_out]``: the ordered bounding ops (see ``layers.bounding.compile_boundings``)
    and, with ``fin = (cols int32, mul f32, add f32)``, the de-normalisation of those columns afterwards."""
    _dev(y, op_col, op_lo, op_hi, op_mul)
    if y.dtype != torch.float32 or not y.is_contiguous():
        raise ValueError("bound_output: y must be contiguous float32")
    n_ops = op_col.numel()
    if op_col.dtype != torch.int32 or op_mul.dtype != torch.int32 or op_lo.dtype != torch.float32 or \
            op_hi.dtype != torch.float32 or op_lo.numel() != n_ops or op_hi.numel() != n_ops or op_mul.numel() != n_ops:
        raise ValueError("bound_output: op lists must be int32 / float32 vectors of one length")
    fc = fm = fa = None
    if fin is not None:
        fc, fm, fa = fin
        _dev(fc, fm, fa)
        if fc.dtype != torch.int32 or fm.dtype != torch.float32 or fa.dtype != torch.float32 or \
                fm.numel() != fc.numel() or fa.numel() != fc.numel():
            raise ValueError("bound_output: fin = (int32 columns, f32 mul, f32 add) of one length")
    v_out = y.shape[-1]
    st = _lib.load().anemoi_bound_output(y.data_ptr(), v_out, y.numel() // max(v_out, 1), n_ops, _ptr(op_col),
                                         _ptr(op_lo), _ptr(op_hi), _ptr(op_mul), 0 if fc is None else fc.numel(),
                                         _ptr(fc), _ptr(fm), _ptr(fa), _stream())
    _lib.check(st, "anemoi_bound_output")
    return y


def advance_input(x: Tensor, y: Tensor, colmap: Tensor, forcing: Optional[Tensor] = None) -> Tensor:
    """Autoregressive input update in place on ``x`` ``[B, T, Ens, G, V_in]`` (f32): shift the time axis by one and
    fill the last slice from the prediction ``y`` ``[B, Ens, G, V_out]`` / the new ``forcing`` ``[B, Ens, G, F]``
    according to ``colmap`` (int32 ``[V_in]``; see include/anemoi_amd.h: anemoi_advance_input)."""
    _dev(x, y, colmap, forcing)
    if x.dtype != torch.float32 or y.dtype != torch.float32 or not x.is_contiguous() or not y.is_contiguous():
        raise ValueError("advance_input: x and y must be contiguous float32")
    b, t, ens, g, v_in = x.shape
    if tuple(y.shape[:3]) != (b, ens, g) or colmap.dtype != torch.int32 or colmap.numel() != v_in:
        raise ValueError(f"advance_input: y {tuple(y.shape)} / colmap do not match x {tuple(x.shape)}")
    f = 0
    if forcing is not None:
        if forcing.dtype != torch.float32 or not forcing.is_contiguous() or tuple(forcing.shape[:3]) != (b, ens, g):
            raise ValueError("advance_input: forcing must be contiguous float32 [B, Ens, G, F]")
        f = forcing.shape[-1]
    st = _lib.load().anemoi_advance_input(x.data_ptr(), b, t, ens, g, v_in, y.data_ptr(), y.shape[-1], _ptr(forcing), f,
                                          colmap.data_ptr(), _stream())
    _lib.check(st, "anemoi_advance_input")
    return x


def prognostic_residual(y: Tensor, x: Tensor, out_idx: Tensor, in_idx: Tensor) -> Tensor:
    """In place: ``y[..., out_idx] += x[:, -1, :, :, in_idx]`` (y f32 ``[B, Ens, G, V_out]`` contiguous)."""
    _dev(y, x, out_idx, in_idx)
    if y.dtype != torch.float32 or not y.is_contiguous():
        raise ValueError("prognostic_residual: y must be contiguous float32")
    b, t, ens, g, v_in = x.shape
    x = x.contiguous().float()
    st = _lib.load().anemoi_prognostic_residual(y.data_ptr(), y.shape[-1], x.data_ptr(), b, t, ens, g, v_in,
                                                out_idx.data_ptr(), in_idx.data_ptr(), out_idx.shape[0], _stream())
    _lib.check(st, "anemoi_prognostic_residual")
    return y


def convert_pad(src: Tensor, dtype: torch.dtype, ld_out: Optional[int] = None) -> Tensor:
    """Copy ``src`` ([rows, cols]) into a ``dtype`` buffer whose rows are zero padded to ``ld_out`` columns."""
    _dev(src)
    _rows(src)
    rows, cols = src.shape
    ld = cols if ld_out is None else ld_out
    out = torch.empty((rows, ld), dtype=dtype, device=src.device)
    if rows == 0:  # (an edge set without edges: torch hands out a null pointer for the empty tensor)
        return out
    st = _lib.load().anemoi_convert_pad(dtype_code(src.dtype), src.data_ptr(), _ld(src), dtype_code(dtype),
                                        out.data_ptr(), ld, rows, cols, _stream())
    _lib.check(st, "anemoi_convert_pad")
    return out


def add(a: Tensor, b: Tensor, out: Optional[Tensor] = None) -> Tensor:
    _dev(a, b, out)
    if a.shape != b.shape or a.dtype != b.dtype:
        raise ValueError("add: shape / dtype mismatch")
    if out is None:
        out = torch.empty(a.shape, dtype=a.dtype, device=a.device)
    st = _lib.load().anemoi_add(dtype_code(a.dtype), a.data_ptr(), _ld(_rows(a)), b.data_ptr(), _ld(_rows(b)),
                                out.data_ptr(), _ld(_rows(out)), a.shape[0], a.shape[1], _stream())
    _lib.check(st, "anemoi_add")
    return out


# ------------------------------------------------------------------------------------------ backward pass, dense half
def transpose(x: Tensor, ld_out: Optional[int] = None) -> Tensor:
    """``x.T`` as a new row-major matrix ``[cols, ld_out]`` (``ld_out >= rows``, extra columns zero): the K-contiguous
    operand of a GEMM that reduces over the rows of ``x``."""
    _dev(x)
    rows, cols = _rows(x).shape
    ld = rows if ld_out is None else ld_out
    out = torch.empty((cols, ld), dtype=x.dtype, device=x.device)
    st = _lib.load().anemoi_transpose(dtype_code(x.dtype), x.data_ptr(), _ld(x), out.data_ptr(), ld, rows, cols, _stream())
    _lib.check(st, "anemoi_transpose")
    return out


def col_sum(x: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """f32 column sums of a row-major matrix (bias gradient), deterministic two-stage reduction."""
    _dev(x)
    rows, cols = _rows(x).shape
    lib = _lib.load()
    if out is None:
        out = torch.empty(cols, dtype=torch.float32, device=x.device)
    elif out.dtype != torch.float32 or out.numel() != cols or not out.is_contiguous():
        raise ValueError("col_sum: out must be a contiguous f32 vector of the column count")
    n_ws = lib.anemoi_col_sum_workspace_floats(rows, cols)
    ws = torch.empty(max(n_ws, 1), dtype=torch.float32, device=x.device)
    st = lib.anemoi_col_sum(dtype_code(x.dtype), x.data_ptr(), _ld(x), rows, cols, out.data_ptr(), ws.data_ptr(), n_ws,
                            _stream())
    _lib.check(st, "anemoi_col_sum")
    return out


def row_dot(a: Tensor, b: Tensor, shift: Optional[Tensor] = None) -> Tensor:
    """``out[r] = sum_c a[r, c] * (b[r, c] - shift[c])`` in f32 (``shift``: optional f32 ``[cols]``)."""
    _dev(a, b, shift)
    rows, cols = _rows(a).shape
    if tuple(_rows(b).shape) != (rows, cols) or a.dtype != b.dtype:
        raise ValueError("row_dot: a and b must have the same shape and dtype")
    if shift is not None and (shift.dtype != torch.float32 or shift.numel() < cols or not shift.is_contiguous()):
        raise ValueError("row_dot: shift must be a contiguous f32 vector with one entry per column")
    out = torch.empty(rows, dtype=torch.float32, device=a.device)
    with _Timed("row_dot", bytes=2 * rows * cols * a.element_size()):
        st = _lib.load().anemoi_row_dot(dtype_code(a.dtype), a.data_ptr(), _ld(_rows(a)), b.data_ptr(), _ld(_rows(b)),
                                        _ptr(shift), out.data_ptr(), rows, cols, _stream())
    _lib.check(st, "anemoi_row_dot")
    return out


def row_scale(x: Tensor, s: Tensor, alpha: float = 1.0, out: Optional[Tensor] = None) -> Tensor:
    """``alpha * s[:, None] * x`` in x's dtype (``s``: contiguous f32 ``[rows]``)."""
    _dev(x, s, out)
    rows, cols = _rows(x).shape
    if s.dtype != torch.float32 or s.numel() != rows or not s.is_contiguous():
        raise ValueError("row_scale: s must be a contiguous f32 vector with one entry per row")
    if out is None:
        out = torch.empty((rows, cols), dtype=x.dtype, device=x.device)
    with _Timed("row_scale", bytes=2 * rows * cols * x.element_size()):
        st = _lib.load().anemoi_row_scale(dtype_code(x.dtype), x.data_ptr(), _ld(_rows(x)), s.data_ptr(), float(alpha),
                                          out.data_ptr(), _ld(_rows(out)), rows, cols, _stream())
    _lib.check(st, "anemoi_row_scale")
    return out


def linear_dual(x: Tensor, w: Tensor, bias: Optional[Tensor], act: str):
    """``(pre, y) = (x @ w.T + bias, act(pre))``: the training forward of Linear + activation.  One launch
    (``anemoi_linear_dual``) for the whole 256-row tiles of a bf16 product and up to 8 rows behind them, the GEMM +
    activation pass pair for a longer remainder."""
    _dev(x, w, bias)
    m, k = _rows(x).shape
    n = w.shape[0]
    if w.dtype != x.dtype or not w.is_contiguous() or w.shape[1] != k:
        raise ValueError("linear_dual: weight must be contiguous [N, K] in the activation dtype, K as x")
    m_main = (m // 256) * 256 if (x.dtype == torch.bfloat16 and n >= 256 and n % 8 == 0 and k >= 128 and k % 64 == 0
                                  and _ld(x) % 8 == 0) else 0
    if m_main < 1024:
        pre = linear(x, w, bias)
        return pre, act_forward(pre, act)
    pre = torch.empty((m, n), dtype=x.dtype, device=x.device)
    y = torch.empty((m, n), dtype=x.dtype, device=x.device)
    if m - m_main <= 8:  # the launch computes up to 8 ragged rows itself (its skinny pass)
        m_main = m
    with _Timed("linear", flops=2 * m_main * n * k, bytes=(m_main * k + n * k + 2 * m_main * n) * 2, m=m_main, n=n, k=k):
        st = _lib.load().anemoi_linear_dual(dtype_code(x.dtype), x.data_ptr(), _ld(x), w.data_ptr(), _ptr(bias),
                                            pre.data_ptr(), n, y.data_ptr(), n, m_main, n, k, _lib.ACT_CODES[act], _stream())
    _lib.check(st, "anemoi_linear_dual")
    if m_main < m:  # the ragged rows
        linear(x[m_main:], w, bias, out=pre[m_main:])
        y[m_main:].copy_(act_forward(pre[m_main:], act))
    return pre, y


def linear_actgrad(x: Tensor, w: Tensor, pre: Tensor, act: str) -> Tensor:
    """``(x @ w.T) * act'(pre)``: the dX GEMM of the Linear behind an activation with the activation's derivative in its
    epilogue (``anemoi_linear_actgrad``; whole 256-row tiles of a bf16 product, GEMM + ``act_backward`` for the rest)."""
    _dev(x, w, pre)
    m, k = _rows(x).shape
    n = w.shape[0]
    if w.dtype != x.dtype or not w.is_contiguous() or w.shape[1] != k or tuple(_rows(pre).shape) != (m, n):
        raise ValueError("linear_actgrad: w must be contiguous [N, K] in the activation dtype, pre [M, N]")
    m_main = (m // 256) * 256 if (x.dtype == torch.bfloat16 and n >= 256 and n % 8 == 0 and k >= 128 and k % 64 == 0
                                  and _ld(x) % 8 == 0 and _ld(_rows(pre)) % 8 == 0) else 0
    if m_main < 1024:
        return act_backward(pre, linear(x, w), act)
    out = torch.empty((m, n), dtype=x.dtype, device=x.device)
    if m - m_main <= 8:  # (as linear_dual)
        m_main = m
    with _Timed("linear", flops=2 * m_main * n * k, bytes=(m_main * k + n * k + 2 * m_main * n) * 2, m=m_main, n=n, k=k):
        st = _lib.load().anemoi_linear_actgrad(dtype_code(x.dtype), x.data_ptr(), _ld(x), w.data_ptr(), pre.data_ptr(),
                                               _ld(_rows(pre)), out.data_ptr(), n, m_main, n, k, _lib.ACT_CODES[act],
                                               _stream())
    _lib.check(st, "anemoi_linear_actgrad")
    if m_main < m:
        out[m_main:].copy_(act_backward(pre[m_main:], linear(x[m_main:], w), act))
    return out


def act_forward(pre: Tensor, act: str, residual: Optional[Tensor] = None) -> Tensor:
    """``act(pre) + residual`` in one pass (the differentiable Linear keeps ``pre`` for the backward)."""
    _dev(pre, residual)
    rows, cols = _rows(pre).shape
    out = torch.empty((rows, cols), dtype=pre.dtype, device=pre.device)
    st = _lib.load().anemoi_act_forward(dtype_code(pre.dtype), _lib.ACT_CODES[act], pre.data_ptr(), _ld(pre),
                                        _ptr(residual), 0 if residual is None else _ld(_rows(residual)), out.data_ptr(),
                                        cols, rows, cols, _stream())
    _lib.check(st, "anemoi_act_forward")
    return out


def act_backward(pre: Tensor, dy: Tensor, act: str) -> Tensor:
    """``dy * act'(pre)`` (``pre`` = the Linear's result before its activation)."""
    _dev(pre, dy)
    rows, cols = _rows(pre).shape
    if tuple(_rows(dy).shape) != (rows, cols) or dy.dtype != pre.dtype:
        raise ValueError("act_backward: pre and dy must have the same shape and dtype")
    out = torch.empty((rows, cols), dtype=pre.dtype, device=pre.device)
    st = _lib.load().anemoi_act_backward(dtype_code(pre.dtype), _lib.ACT_CODES[act], pre.data_ptr(), _ld(pre),
                                         dy.data_ptr(), _ld(dy), out.data_ptr(), cols, rows, cols, _stream())
    _lib.check(st, "anemoi_act_backward")
    return out


def layer_norm_backward(x: Tensor, stats: Tensor, gamma: Tensor, dy: Tensor, dres: Optional[Tensor] = None):
    """``(dx, dgamma, dbeta)`` of ``layer_norm(x) * gamma + beta`` from the forward's ``row_stats(x)``; ``dres`` (optional,
    x's shape and dtype) is added to ``dx`` in the same pass (the skip connection's gradient)."""
    _dev(x, stats, gamma, dy, dres)
    rows, c = _rows(x).shape
    if tuple(_rows(dy).shape) != (rows, c) or dy.dtype != x.dtype or stats.shape != (rows, 2):
        raise ValueError("layer_norm_backward: shapes of x, dy, stats do not match")
    lib = _lib.load()
    dx = torch.empty((rows, c), dtype=x.dtype, device=x.device)
    dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(c, dtype=torch.float32, device=x.device)
    n_ws = lib.anemoi_layer_norm_backward_workspace_floats(rows, c)
    ws = torch.empty(n_ws, dtype=torch.float32, device=x.device)
    gamma = gamma.detach().float().contiguous()
    if dres is not None and (tuple(_rows(dres).shape) != (rows, c) or dres.dtype != x.dtype):
        raise ValueError("layer_norm_backward: dres must have x's shape and dtype")
    st = lib.anemoi_layer_norm_backward(dtype_code(x.dtype), x.data_ptr(), _ld(x), stats.data_ptr(), gamma.data_ptr(),
                                        dy.data_ptr(), _ld(dy), _ptr(dres), 0 if dres is None else _ld(_rows(dres)),
                                        dx.data_ptr(), c, rows, c, dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(),
                                        n_ws, _stream())
    _lib.check(st, "anemoi_layer_norm_backward")
    return dx, dgamma, dbeta


def weight_grad(dpre: Tensor, x: Tensor, k: int, want_bias: bool = False, transposed_route: bool = False,
                out: Optional[Tensor] = None):
    """``dW [N, k] = dpre^T @ x[:, :k]`` in f32 (``dpre [M, N]``, ``x [M, >= k]`` in the compute dtype): the reduction
    over the M rows is cut into chunks -- chunked transposes, one batched GEMM on the 128 x 128 kernel, a deterministic
    sum of the partial results -- so that a small ``[N, k]`` result still fills the chip (a 1024 x 192 gradient over
    542 080 rows took 7 ms on 16 workgroups without the split).  ``want_bias``: returns ``(dW, db)`` with
    ``db = dpre.sum(0)`` (f32); for bf16 the column sums come out of the transpose of ``dpre`` (per-tile partials), not
    out of a second pass over it.  ``transposed_route=True`` (tests, tools/dw_bench.py) takes the transposes + NT route even
    where the TN kernel applies (bf16, 16-byte aligned operands, M >= 128), which otherwise runs.  ``out`` (optional, f32
    ``[N * k (+ N)]`` contiguous): where the TN route leaves ``dW`` (and ``db`` behind it) -- the results are then views of
    it (a caller's stacked gradient buffer, autograd.GradSink); ignored by the other route."""
    _dev(dpre, x)
    m, n = _rows(dpre).shape
    kmul = k_multiple(dpre.dtype)
    # bf16: partial results in bf16 from the persistent 256 x 256 kernel (f32 accumulation inside, as an autocast matmul
    # rounds them), summed in f32; f32: the exact 128 x 128 kernel
    fast = dpre.dtype == torch.bfloat16 and k % 8 == 0
    xr = _rows(x)
    if (fast and n % 8 == 0 and _ld(dpre) % 8 == 0 and _ld(xr) % 8 == 0 and dpre.data_ptr() % 16 == 0
            and xr.data_ptr() % 16 == 0 and m >= 128 and not transposed_route):
        # no transposed copies: the TN kernel reads dpre and x as they lie (ds_read_b64_tr_b16 fragments), f32 partials
        tiles = ((n + 255) // 256) * ((k + 255) // 256)
        chunks = max(1, min(256 // tiles if tiles <= 256 else 1, m // 2048))
        ld_max = max(_ld(dpre), _ld(xr))
        row_cap = ((1 << 31) - 1) // (2 * ld_max) // 64 * 64  # descriptor range of one chunk
        chunk_rows = max(128, min(round_up((m + chunks - 1) // chunks, 64), row_cap))
        chunks = (m + chunk_rows - 1) // chunk_rows
        # one buffer [chunks, n * k (+ n)]: the weight partials and, behind them, the bias partials of a chunk -- ONE column
        # sum over the chunks finishes both
        width = n * k + (n if want_bias else 0)
        if out is not None and (out.dtype != torch.float32 or out.numel() != width or not out.is_contiguous()):
            raise ValueError(f"weight_grad: out must be a contiguous f32 vector of {width} elements")
        if out is not None and chunks == 1:
            part = out.view(1, width)
        else:
            part = torch.empty((chunks, width), dtype=torch.float32, device=dpre.device)
        st = _lib.load().anemoi_weight_grad_tn(dpre.data_ptr(), _ld(dpre), xr.data_ptr(), _ld(xr), part.data_ptr(), width,
                                               part[:, n * k:].data_ptr() if want_bias else None, width, m, n, k, chunk_rows,
                                               _stream())
        _lib.check(st, "anemoi_weight_grad_tn")
        total = part[0] if chunks == 1 else col_sum(part, out=None if out is None else out.view(width))
        dw = total[: n * k].view(n, k)
        return (dw, total[n * k:]) if want_bias else dw
    tile = 256 if fast else 128
    tiles = ((n + tile - 1) // tile) * ((k + tile - 1) // tile)
    chunks = max(1, min(256 // tiles if tiles <= 256 else 1, m // 2048))
    chunk_rows = round_up((m + chunks - 1) // chunks, max(kmul, 128 if fast else kmul))
    chunks = (m + chunk_rows - 1) // chunk_rows
    lib = _lib.load()
    code = dtype_code(dpre.dtype)
    at = torch.empty((chunks, n, chunk_rows), dtype=dpre.dtype, device=dpre.device)
    bt = torch.empty((chunks, k, chunk_rows), dtype=dpre.dtype, device=dpre.device)
    partial = None
    if want_bias and fast and n % 4 == 0:
        partial = torch.empty((lib.anemoi_transpose_colsum_rows(m, chunk_rows), n), dtype=torch.float32, device=dpre.device)
    st = lib.anemoi_transpose_chunked(code, dpre.data_ptr(), _ld(dpre), at.data_ptr(), chunk_rows, m, n, chunk_rows,
                                      _ptr(partial), _stream())
    _lib.check(st, "anemoi_transpose_chunked")
    st = lib.anemoi_transpose_chunked(code, x.data_ptr(), _ld(_rows(x)), bt.data_ptr(), chunk_rows, m, k, chunk_rows, None,
                                      _stream())
    _lib.check(st, "anemoi_transpose_chunked")
    part = torch.empty((chunks, n, k), dtype=dpre.dtype if fast else torch.float32, device=dpre.device)
    st = lib.anemoi_linear_batched(code, dtype_code(part.dtype), at.data_ptr(), chunk_rows, n * chunk_rows, bt.data_ptr(),
                                   k * chunk_rows, part.data_ptr(), k, n * k, chunks, n, k, chunk_rows, _stream())
    _lib.check(st, "anemoi_linear_batched")
    if part.dtype != torch.float32 and chunks == 1:
        dw = part[0].float()
    elif chunks == 1:
        dw = part[0]
    else:
        dw = col_sum(part.view(chunks, n * k)).view(n, k)
    if not want_bias:
        return dw
    return dw, (col_sum(partial) if partial is not None else col_sum(dpre))
