"""Autograd functions of the differentiable route: forward AND backward on the HIP kernels.

Dense half
* ``linear(x, weight, bias, act, residual)``: forward = the fused GEMM (with an activation: ``anemoi_linear_dual``, the
  pre-activation as a second output); backward ``dX = dpre W`` and ``dW = dpre^T X`` (f32) on the same GEMM kernels over
  operands from the register transposes, ``db`` from the per-tile column sums those transposes leave behind;
* ``mlp2(x, w1, b1, w2, b2, act, residual)``: Linear -> act -> Linear (+ residual) as ONE node; the second Linear's dX
  GEMM applies ``act'(pre)`` in its epilogue (``anemoi_linear_actgrad``);
* ``layer_norm``: ``anemoi_layer_norm_backward`` from the forward's row statistics.
Edge half (``csrc/edge_backward.hip``)
* ``gt_edge_attention`` / ``gt_edge_attention_packed`` / the processor block's ``_GTEdgeAttentionSelf``: the folded edge
  phase; the forward leaves the softmax normaliser, the backward is one batched sweep per direction;
* ``gt_conv``: the conv on explicit per-edge features (any edge_dim);
* ``gather_add_act`` / ``segment_sum`` (GNN), ``mhsa`` (+ attention dropout), ``permute_rows``.
``gt_processor_block`` / ``gt_mapper_block`` compose them exactly as the inference path composes its launches; the
``lin_edge`` fold is torch algebra on the parameters, so autograd carries gradients back to ``lin_edge`` / ``lin_query`` /
``projection``.  What torch derives for the reference's ``nn.Linear`` / ``nn.GELU`` / ``nn.LayerNorm`` /
``GraphTransformerConv`` (layers/block.py:504-508, 602-635, layers/conv.py:98-142, layers/mlp.py:74-84) when
anemoi-training calls ``.backward()``.  Parameters and their gradients stay f32; activations and activation gradients
are in the compute dtype.  No atomics anywhere: gradients are reproducible bit for bit.
"""

from __future__ import annotations

import os
from typing import Optional

import torch
from torch import Tensor

from . import ops


def _pack(w: Tensor, dtype: torch.dtype, k_pad: Optional[int] = None) -> Tensor:
    """``[N, K]`` f32 parameter -> compute dtype with K zero-padded to the GEMM's slab multiple (or to ``k_pad`` columns)."""
    n, k = w.shape
    kp = ops.round_up(k, ops.k_multiple(dtype)) if k_pad is None else k_pad
    if kp == k:  # no padding: the cast alone (one launch; nothing for a contiguous weight in the compute dtype)
        return w.detach().to(dtype).contiguous()
    out = torch.zeros((n, kp), dtype=dtype, device=w.device)
    out[:, :k] = w.detach().to(dtype)
    return out


class GradSink:
    """``[count, N * K (+ N)]`` f32: where the weight (and bias) gradients of the SAME Linear of ``count`` equally shaped
    blocks land -- slot ``i`` is the ``out`` of block i's weight-gradient reduction (``ops.weight_grad``), so the gradient of
    the stacked weight ``[count, N, K]`` is complete without a copy when the last block has run (:class:`_Unstack`).
    Allocated on first use in a backward, released when every part has been handed to autograd."""

    def __init__(self, count: int, n: int, k: int, bias: bool, device, stacked_parts: int = 1) -> None:
        self.count, self.n, self.k, self.bias, self.device = count, n, k, bias, device
        self.width = n * k + (n if bias else 0)
        self.buf: Optional[Tensor] = None
        self.stacked_parts = stacked_parts  # how many parts ("w", "b") an _Unstack collects before the buffer is let go
        # a bias part that no _Unstack collects goes to a LEAF parameter, whose AccumulateGrad would keep the 1-D view -- and
        # with it the whole [count, N * K + N] buffer -- alive as ``.grad``: such a bias gradient is handed out as a copy
        self.copy_bias = bias and stacked_parts < 2
        self._pending = 0

    def slot(self, i: int) -> Tensor:
        if self.buf is None:
            self.buf = torch.empty((self.count, self.width), dtype=torch.float32, device=self.device)
            self._pending = self.stacked_parts
        return self.buf[i]

    def stacked(self, part: str, grads) -> Optional[Tensor]:
        """The stacked gradient of ``part`` ("w" / "b") if every block's gradient IS its slot of the buffer, else None."""
        buf = self.buf
        if buf is None:
            return None
        # this part is collected now, whichever way: when the last one is, the sink lets go of its buffer, so that a second
        # backward over a retained graph gets fresh storage instead of slots that gradients already handed out still alias
        self._pending -= 1
        if self._pending <= 0:
            self.buf = None
        if len(grads) != self.count:
            return None
        off = 0 if part == "w" else self.n * self.k
        size = self.n * self.k if part == "w" else self.n
        esz = buf.element_size()
        for i, g in enumerate(grads):
            if (g is None or g.dtype != buf.dtype or g.numel() != size or not g.is_contiguous()
                    or g.data_ptr() != buf.data_ptr() + (i * self.width + off) * esz):
                return None
        out = buf[:, off:off + size]
        return out.view(self.count, self.n, self.k) if part == "w" else out


class WeightPrep:
    """One Linear of one block, prepared by its processor for a whole forward + backward: ``w`` the weight in the compute
    dtype (K padded), ``wt`` its transpose ``[K, N]`` (the dX GEMM's operand) or None, ``sink`` / ``index`` the slot of a
    :class:`GradSink` its weight gradient is reduced into (or None)."""

    __slots__ = ("w", "wt", "sink", "index")

    def __init__(self, w: Tensor, wt: Optional[Tensor] = None, sink: Optional[GradSink] = None, index: int = 0) -> None:
        self.w, self.wt, self.sink, self.index = w, wt, sink, index

    def grad_out(self, has_bias: bool, want_w: bool, want_b: bool) -> Optional[Tensor]:
        s = self.sink
        if s is None or not want_w or s.bias != (has_bias and want_b):
            return None
        return s.slot(self.index)


class _Unstack(torch.autograd.Function):
    """``stacked [L, ...] -> L`` tensors (``unbind``).  Backward: when the L gradients are the slots of ``sink`` they ARE the
    stacked gradient already -- no ``torch.stack`` over ``L x N x K`` floats (config 3: 0.9 GB per step)."""

    @staticmethod
    def forward(ctx, stacked: Tensor, sink: Optional[GradSink], part: str):
        ctx.sink, ctx.part, ctx.shape = sink, part, stacked.shape
        ctx.set_materialize_grads(False)
        return tuple(t.view_as(t) for t in stacked.unbind(0))

    @staticmethod
    def backward(ctx, *grads):
        if all(g is None for g in grads):
            return None, None, None
        got = ctx.sink.stacked(ctx.part, grads) if ctx.sink is not None else None
        if got is None:
            ref = next(g for g in grads if g is not None)
            got = torch.stack([torch.zeros_like(ref) if g is None else g for g in grads])
        return got.reshape(ctx.shape), None, None


_TORCH_ACT = {"GELU": torch.nn.functional.gelu, "SiLU": torch.nn.functional.silu, "ReLU": torch.relu}


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias: Optional[Tensor], act: str, residual: Optional[Tensor],
                prep: Optional[WeightPrep] = None, padded_input: bool = False):
        dtype = x.dtype
        k = weight.shape[1]
        kp = ops.round_up(k, ops.k_multiple(dtype))
        if padded_input and prep is None and x.shape[1] == ops.round_up(k + 1, ops.k_multiple(dtype)):
            # ``[features | 1 | 0-pad]`` rows (training._AssembleNodes): the constant column and the padding behind it meet
            # zero weights.  Opt-in and for exactly that width -- any other width is a mis-wired call and raises below
            kp = x.shape[1]
        if x.shape[1] not in (k, kp):
            raise ValueError(f"linear: x has {x.shape[1]} columns, weight expects {k}")
        xk = x if x.shape[1] == kp else ops.convert_pad(x, dtype, kp)
        w = _pack(weight, dtype, kp) if prep is None else prep.w
        ctx.prep = prep
        b = None if bias is None else bias.detach().float().contiguous()
        if act == "Identity":
            pre = None
            y = ops.linear(xk, w, b, residual=residual)
        else:
            # the pre-activation is an output of its own here (the backward needs act'(pre)); the activation and the
            # residual then run on the stored value instead of in the GEMM epilogue
            if residual is None and act in ("GELU", "SiLU", "ReLU") and w.shape[0] % (16 // w.element_size()) == 0:
                pre, y = ops.linear_dual(xk, w, b, act)  # one launch: the pre-activation is a second output
                ctx.save_for_backward(xk, weight, pre)
                ctx.act, ctx.has_bias, ctx.has_res, ctx.k, ctx.x_cols = act, bias is not None, False, k, x.shape[1]
                return y
            pre = ops.linear(xk, w, b)
            if pre.shape[1] % (16 // pre.element_size()) == 0:
                y = ops.act_forward(pre, act, residual)
            else:
                y = _TORCH_ACT[act](pre.float()).to(dtype)
                if residual is not None:
                    y = ops.add(y, residual)
        ctx.save_for_backward(xk, weight, pre)
        ctx.act, ctx.has_bias, ctx.has_res, ctx.k, ctx.x_cols = act, bias is not None, residual is not None, k, x.shape[1]
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        xk, weight, pre = ctx.saved_tensors
        dtype = xk.dtype
        dy = dy.contiguous()
        dpre = dy if ctx.act == "Identity" else ops.act_backward(pre, dy, ctx.act)
        n, k = weight.shape
        kmul = ops.k_multiple(dtype)
        prep = ctx.prep
        dx = dw = db = dres = None
        if ctx.needs_input_grad[0]:
            # dX [M, K] = dpre [M, N] @ W [N, K]: a Linear whose weight is W^T [K, N] (N is the reduction dimension)
            np_ = ops.round_up(n, kmul)
            if prep is not None and prep.wt is not None:
                wt = prep.wt
            else:
                wt = ops.transpose(weight.detach().to(dtype).contiguous(), ld_out=np_)
            dp = dpre if n == np_ else ops.convert_pad(dpre, dtype, np_)
            if (dtype == torch.bfloat16 and wt.shape[0] < 256 and dp.shape[0] >= 65536 and np_ >= 128
                    and os.environ.get("ANEMOI_AMD_TRAIN_WIDE_DX", "1") != "0"):
                # few input features on very many rows (the mappers' embeddings of the grid nodes: 542 080 x 1024 -> 192 at
                # config 3): fewer than 256 output columns would take the 128 x 128 kernel (0.49 ms, 2.3 TB/s of the dpre
                # it reads); with the transposed weight padded to 256 zero rows the persistent kernel runs it
                wide = torch.zeros((256, np_), dtype=dtype, device=wt.device)
                wide[: wt.shape[0]] = wt
                dx = ops.linear(dp, wide)[:, : wt.shape[0]]
            else:
                dx = ops.linear(dp, wt)
            if ctx.x_cols != k:
                dx = ops.convert_pad(dx, dtype, ctx.x_cols)
        if ctx.needs_input_grad[1]:
            # dW [N, K] = dpre^T [N, M] @ X [M, K]: a Linear with x' = dpre^T, weight' = X^T, reduction over the M rows
            want_b = ctx.has_bias and ctx.needs_input_grad[2]
            out = None if prep is None or weight.dtype != torch.float32 else prep.grad_out(ctx.has_bias, True, want_b)
            if want_b:
                dw, db = ops.weight_grad(dpre, xk, k, want_bias=True, out=out)  # the bias gradient rides on dpre's transpose
                dw = dw.to(weight.dtype)
                if out is not None and prep.sink.copy_bias:
                    db = db.clone()  # N floats; see GradSink.copy_bias
            else:
                dw = ops.weight_grad(dpre, xk, k, out=out).to(weight.dtype)
        if db is None and ctx.has_bias and ctx.needs_input_grad[2]:
            db = ops.col_sum(dpre)
        if ctx.has_res and ctx.needs_input_grad[4]:
            dres = dy
        return dx, dw, db, None, dres, None, None


class _MLP2(torch.autograd.Function):
    """``Linear2(act(Linear1(x))) + residual`` as ONE node of the autograd graph (the node MLPs of every block,
    reference layers/block.py:504-508, layers/mlp.py:74-84).  Forward: the pre-activation is a second output of the first
    GEMM (``ops.linear_dual``), the residual rides in the second GEMM's epilogue.  Backward: the dX GEMM of Linear2
    multiplies by ``act'(pre)`` in its epilogue (``ops.linear_actgrad``) and so delivers the gradient of Linear1's
    pre-activation directly; bias gradients come from the weight-gradient transposes.  bf16 with slab-multiple widths;
    :func:`mlp2` composes two :class:`_Linear` nodes otherwise."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, act: str, residual, prep1: Optional[WeightPrep] = None,
                prep2: Optional[WeightPrep] = None):
        dtype = x.dtype
        w1p = _pack(w1, dtype) if prep1 is None else prep1.w
        w2p = _pack(w2, dtype) if prep2 is None else prep2.w
        ctx.prep1, ctx.prep2 = prep1, prep2
        pre, h = ops.linear_dual(x, w1p, None if b1 is None else b1.detach().float().contiguous(), act)
        y = ops.linear(h, w2p, None if b2 is None else b2.detach().float().contiguous(), residual=residual)
        ctx.save_for_backward(x, w1, w2, pre, h)
        ctx.act, ctx.has_b1, ctx.has_b2, ctx.has_res = act, b1 is not None, b2 is not None, residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w1, w2, pre, h = ctx.saved_tensors
        dtype = x.dtype
        dy = dy.contiguous()
        need = ctx.needs_input_grad
        prep1, prep2 = ctx.prep1, ctx.prep2

        def sink_out(prep, w, has_b, want_b):
            return None if prep is None or w.dtype != torch.float32 else prep.grad_out(has_b, True, want_b)

        def transposed(prep, w):
            return prep.wt if prep is not None and prep.wt is not None else ops.transpose(w.detach().to(dtype).contiguous())

        dx = dw1 = db1 = dw2 = db2 = None
        if need[3]:
            want_b2 = ctx.has_b2 and need[4]
            out2 = sink_out(prep2, w2, ctx.has_b2, want_b2)
            if want_b2:
                dw2, db2 = ops.weight_grad(dy, h, w2.shape[1], want_bias=True, out=out2)
            else:
                dw2 = ops.weight_grad(dy, h, w2.shape[1], out=out2)
            dw2 = dw2.to(w2.dtype)
        if db2 is None and ctx.has_b2 and need[4]:
            db2 = ops.col_sum(dy)
        if need[0] or need[1] or (ctx.has_b1 and need[2]):
            # d pre = (dy W2) * act'(pre): Linear2's dX GEMM with the activation's derivative in its epilogue
            dpre = ops.linear_actgrad(dy, transposed(prep2, w2), pre, ctx.act)
            if need[1]:
                want_b1 = ctx.has_b1 and need[2]
                out1 = sink_out(prep1, w1, ctx.has_b1, want_b1)
                if want_b1:
                    dw1, db1 = ops.weight_grad(dpre, x, w1.shape[1], want_bias=True, out=out1)
                else:
                    dw1 = ops.weight_grad(dpre, x, w1.shape[1], out=out1)
                dw1 = dw1.to(w1.dtype)
            if db1 is None and ctx.has_b1 and need[2]:
                db1 = ops.col_sum(dpre)
            if need[0]:
                dx = ops.linear(dpre, transposed(prep1, w1))
        return dx, dw1, db1, dw2, db2, None, (dy if ctx.has_res and need[6] else None), None, None


def mlp2(x: Tensor, w1: Tensor, b1: Optional[Tensor], w2: Tensor, b2: Optional[Tensor], act: str,
         residual: Optional[Tensor] = None, prep1: Optional[WeightPrep] = None,
         prep2: Optional[WeightPrep] = None) -> Tensor:
    """``linear(linear(x, w1, b1, act), w2, b2, residual=residual)`` -- fused into one autograd node (:class:`_MLP2`) when
    the shapes allow the fused GEMM epilogues (bf16, every width a multiple of the 64-element K slab)."""
    km = ops.k_multiple(x.dtype)
    fused = (x.dtype == torch.bfloat16 and act in ("GELU", "SiLU", "ReLU") and x.shape[1] == w1.shape[1]
             and w1.shape[1] % km == 0 and w1.shape[0] % km == 0 and w2.shape[0] % km == 0 and w2.shape[1] == w1.shape[0]
             and x.shape[0] >= 1024 and w1.shape[0] >= 256 and w2.shape[0] >= 256 and x.is_contiguous())
    if not fused:
        return linear(linear(x, w1, b1, act, prep=prep1), w2, b2, "Identity", residual, prep=prep2)
    return _MLP2.apply(x, w1, b1, w2, b2, act, residual, prep1, prep2)


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, gamma: Tensor, beta: Tensor, eps: float):
        y, stats = ops.layer_norm_with_stats(x, gamma.detach().float().contiguous(), beta.detach().float().contiguous(),
                                             eps)
        ctx.save_for_backward(x, stats, gamma)
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        x, stats, gamma = ctx.saved_tensors
        dx, dgamma, dbeta = ops.layer_norm_backward(x, stats, gamma, dy.contiguous())
        return dx, dgamma.to(gamma.dtype), dbeta.to(gamma.dtype), None


class _LayerNormSkip(torch.autograd.Function):
    """``(LayerNorm(x), x)``: the normalised rows for the branch AND the rows themselves for the skip connection around
    it (``x + f(norm(x))``, reference layers/block.py:504-508, 614-635) as ONE autograd node -- the backward receives both
    gradients and ``anemoi_layer_norm_backward`` adds the skip's into ``dx`` in its own pass, instead of autograd
    accumulating the two contributions with a separate add over ``[N, C]`` (36 of them per config-3 training step)."""

    @staticmethod
    def forward(ctx, x: Tensor, gamma: Tensor, beta: Tensor, eps: float):
        y, stats = ops.layer_norm_with_stats(x, gamma.detach().float().contiguous(), beta.detach().float().contiguous(),
                                             eps)
        ctx.save_for_backward(x, stats, gamma)
        ctx.set_materialize_grads(False)  # an unused output arrives as None below, not as a zero matrix to normalise
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy: Optional[Tensor], dskip: Optional[Tensor]):
        x, stats, gamma = ctx.saved_tensors
        if dy is None:  # the branch is unused: only the skip carries a gradient
            return dskip, None, None, None
        dx, dgamma, dbeta = ops.layer_norm_backward(x, stats, gamma, dy.contiguous(),
                                                    None if dskip is None else dskip.contiguous())
        return dx, dgamma.to(gamma.dtype), dbeta.to(gamma.dtype), None


def layer_norm_skip(x: Tensor, gamma: Tensor, beta: Tensor, eps: float = 1e-5):
    """``(layer_norm(x), x)`` -- use the second result for the skip connection around the normalised branch."""
    return _LayerNormSkip.apply(x, gamma, beta, eps)


def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None, act: str = "Identity",
           residual: Optional[Tensor] = None, prep: Optional[WeightPrep] = None, padded_input: bool = False) -> Tensor:
    """``act(x @ weight.T + bias) + residual`` with gradients for ``x``, ``weight``, ``bias`` and ``residual``.
    ``x`` / ``residual`` in the compute dtype (f32 or bf16), ``weight [N, K]`` / ``bias [N]`` f32 parameters.  ``act``
    outside the kernel's epilogues (Identity / GELU / SiLU / ReLU; the reference takes any ``torch.nn`` activation by
    name): the product runs here, the activation as a torch module behind it.  ``x`` must have K columns (or K rounded up
    to the slab); ``padded_input=True`` additionally admits ``[features | 1 | 0-pad]`` rows of exactly
    ``round_up(K + 1, slab)`` columns (``training._AssembleNodes``); every other width raises ``ValueError``."""
    if act not in ("Identity", "GELU", "SiLU", "ReLU"):
        y = getattr(torch.nn, act)()(_Linear.apply(x, weight, bias, "Identity", None, prep, padded_input))
        return y if residual is None else y + residual
    return _Linear.apply(x, weight, bias, act, residual, prep, padded_input)


def layer_norm(x: Tensor, gamma: Tensor, beta: Tensor, eps: float = 1e-5) -> Tensor:
    """``LayerNorm(x) * gamma + beta`` over the last dimension with gradients for ``x``, ``gamma``, ``beta``."""
    return _LayerNorm.apply(x, gamma, beta, eps)


class _ScaledLinear(torch.autograd.Function):
    """``y = s[:, None] * (x @ f.T) + b``: ``x [M, Kp]`` in the compute dtype (K padded), ``f [N, Kp]`` / ``b [N]`` f32,
    ``s [M]`` f32 -- the product behind a LayerNorm whose mean has been removed algebraically (``s`` = the row's rstd).
    Forward on ``anemoi_linear_ln`` (statistics ``{s, 0}``, zero column sums); backward: ``df = dy^T (s x)`` (TN weight
    gradient on the scaled K-narrow rows), ``g = dy f`` (one K = N product onto the Kp narrow columns), ``dx = s g``,
    ``ds = sum_k g x`` (``anemoi_row_dot`` over the Kp columns: ``sum_n dy (x f^T)_n`` regrouped, so neither the wide
    output nor the unscaled product is read again), ``db = sum_m dy``."""

    @staticmethod
    def forward(ctx, x: Tensor, f: Tensor, s: Tensor, b: Tensor):
        dtype = x.dtype
        fp = f.detach().to(dtype).contiguous()
        sd = s.detach().float()
        stats = torch.stack([sd, torch.zeros_like(sd)], dim=1).contiguous()
        zeros = torch.zeros(f.shape[0], dtype=torch.float32, device=x.device)
        y = ops.linear(x, fp, b.detach().float().contiguous(), ln=(stats, zeros))
        ctx.save_for_backward(x, f, s, b)
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        x, f, s, b = ctx.saved_tensors
        dtype = x.dtype
        dy = dy.contiguous()
        need = ctx.needs_input_grad
        n, kp = f.shape
        sd = s.detach().float().contiguous()
        dx = df = ds = db = None
        if need[0] or need[2]:
            np_ = ops.round_up(n, ops.k_multiple(dtype))
            ft = ops.transpose(f.detach().to(dtype).contiguous(), ld_out=np_)
            dp = dy if n == np_ else ops.convert_pad(dy, dtype, np_)
            g = ops.linear(dp, ft)
            if need[2]:
                ds = ops.row_dot(g, x)
            if need[0]:
                dx = ops.row_scale(g, sd, out=g)
        if need[1]:
            xs = ops.row_scale(x, sd)
            if need[3]:
                df, db = ops.weight_grad(dy, xs, kp, want_bias=True)
            else:
                df = ops.weight_grad(dy, xs, kp)
            df = df.to(f.dtype)
        if db is None and need[3]:
            db = ops.col_sum(dy)
        return dx, df, ds, (None if db is None else db.to(b.dtype))


class _QuadForm(torch.autograd.Function):
    """``q[i] = x_i^T M x_i`` for a SYMMETRIC ``M [K, K]`` (f32), ``x [rows, K]`` in the compute dtype: one K x K product on
    the rows + one row dot (f32 sums of the bf16 products).  Backward: ``dx = 2 g (x M)`` (the saved product, row scaled),
    ``dM = x^T (g x)`` (TN weight gradient)."""

    @staticmethod
    def forward(ctx, x: Tensor, m: Tensor):
        z = ops.linear(x, m.detach().to(x.dtype).contiguous())
        ctx.save_for_backward(x, z)
        return ops.row_dot(z, x)

    @staticmethod
    def backward(ctx, g: Tensor):
        x, z = ctx.saved_tensors
        g = g.detach().float().contiguous()
        dx = ops.row_scale(z, g, 2.0) if ctx.needs_input_grad[0] else None
        dm = ops.weight_grad(ops.row_scale(x, g), x, x.shape[1]) if ctx.needs_input_grad[1] else None
        return dx, dm


def folded_embedding_ln_linear(x: Tensor, emb_w: Tensor, emb_b: Optional[Tensor], gamma: Tensor, beta: Tensor, eps: float,
                               w_rows: Tensor, bias: Optional[Tensor], augmented: bool = False) -> Tensor:
    """``Linear(LayerNorm(emb(x)))`` on the RAW node features (mapper embedding -> block LayerNorm -> k|v or x_r|q|u
    Linear, reference layers/mapper.py:322-331 + layers/block.py:516-528), differentiable: the training form of
    ``runtime.fold_embedded_layer_norm``.  With the channel-centred embedding ``A = [E_c | b_c]`` (the LayerNorm's mean is
    gone algebraically) and ``x_aug = [x | 1 | 0-pad]``:

        var_i = x_aug_i^T (A^T A / C) x_aug_i,   y_i = rsqrt(var_i + eps) * (x_aug_i F^T) + b',
        F = (W * gamma) A,   b' = b + W beta

    -- a K = k_in + 1 product instead of the embedding GEMM's successor at K = C, and no ``[M, C]`` LayerNorm pass.  The
    algebra on the parameters is plain torch (autograd carries the gradients to the embedding, the LayerNorm and the
    Linear), the two products on the rows are autograd nodes on the HIP kernels.  ``augmented``: ``x`` IS ``x_aug`` already
    (``[M, kp]`` with the constant 1 in column ``k_in`` and zeros behind it: ``training._AssembleNodes`` writes the model
    input that way), so no concatenation pass over the rows."""
    dtype = x.dtype
    c, k_in = emb_w.shape
    if x.shape[1] < k_in:
        raise ValueError(f"folded_embedding_ln_linear: x has {x.shape[1]} columns, the embedding expects {k_in}")
    kp = ops.round_up(k_in + 1, ops.k_multiple(dtype))
    m = x.shape[0]
    if augmented:
        if x.shape[1] != kp:
            raise ValueError(f"folded_embedding_ln_linear: augmented rows need {kp} columns, got {x.shape[1]}")
        xa = x
    else:
        xa = torch.cat([x[:, :k_in], torch.ones((m, 1), dtype=dtype, device=x.device),
                        torch.zeros((m, kp - k_in - 1), dtype=dtype, device=x.device)], dim=1)
    e_c = emb_w.float() - emb_w.float().mean(0, keepdim=True)
    b_e = torch.zeros(c, dtype=torch.float32, device=x.device) if emb_b is None else emb_b.float()
    b_c = b_e - b_e.mean()
    a = torch.cat([e_c, b_c[:, None], torch.zeros((c, kp - k_in - 1), dtype=torch.float32, device=x.device)], dim=1)
    var = _QuadForm.apply(xa, a.t() @ a / c)
    s = torch.rsqrt(var.clamp_min(0.0) + eps)
    w = w_rows.float()
    f = (w * gamma.float()[None, :]) @ a
    b2 = w @ beta.float()
    if bias is not None:
        b2 = b2 + bias.float()
    return _ScaledLinear.apply(xa, f, s, b2)


# ------------------------------------------------------------------------------------------ edge phase
def _transposed_csr(plan):
    """(rowptr_t, eid_t, dst_t) of the source-major view of a destination-sorted plan, cached on the plan."""
    cached = getattr(plan, "_transposed", None)
    if cached is None:
        col = plan.col.long()
        counts = (plan.rowptr[1:] - plan.rowptr[:-1]).long()
        dst_of_edge = torch.repeat_interleave(torch.arange(plan.n_dst, device=col.device), counts)
        order = torch.argsort(col, stable=True)
        rowptr_t = torch.zeros(plan.n_src + 1, dtype=torch.int64, device=col.device)
        torch.cumsum(torch.bincount(col, minlength=plan.n_src), 0, out=rowptr_t[1:])
        cached = (rowptr_t.to(torch.int32), order.to(torch.int32).contiguous(),
                  dst_of_edge[order].to(torch.int32).contiguous(), dst_of_edge.to(torch.int32).contiguous())
        plan._transposed = cached
    return cached


def _edge_backward(q, k, v, dout, u, dt, lse, edge_attr, plan, h: int, up: int, dq, dk, dv, du, want_dattr: bool,
                   dxr: Optional[Tensor] = None):
    """The three backward kernels of the folded edge phase (csrc/edge_backward.hip) on column views: ``q, dout, u, dt``
    ``[n_dst, .]``, ``k, v`` ``[n_src, C]`` in, ``dq, du`` ``[n_dst, .]`` and ``dk, dv`` ``[n_src, C]`` out (any row
    pitch: the processor block hands in the column ranges of ONE ``d(x_r|q|k|v|u)`` buffer).  ``dxr`` (optional,
    ``[n_dst, C]`` view): receives ``dout`` (the self term's gradient) in the destination sweep.  Returns d edge_attr."""
    from . import _lib

    n_dst, c = q.shape
    n_src = k.shape[0]
    n_edges = plan.col.shape[0]
    dev = q.device
    alpha = torch.empty((n_edges, h), dtype=torch.float32, device=dev)
    w = torch.empty((n_edges, h), dtype=torch.float32, device=dev)
    dsum = torch.empty((n_dst, h), dtype=torch.float32, device=dev)
    lib, code, stream = _lib.load(), ops.dtype_code(q.dtype), ops._stream()
    ld = lambda t: ops._ld(ops._rows(t))  # noqa: E731
    if ld(k) != ld(v) or ld(dk) != ld(dv):
        raise ValueError("gt_edge_attention: k / v (and dk / dv) must share their leading dimension")
    st = lib.anemoi_gt_edge_attention_folded_backward_dst(
        code, q.data_ptr(), ld(q), k.data_ptr(), v.data_ptr(), ld(k), dout.data_ptr(), ld(dout), u.data_ptr(), ld(u),
        dt.data_ptr(), ld(dt), lse.data_ptr(), edge_attr.data_ptr(), up, plan.rowptr.data_ptr(), plan.col.data_ptr(),
        alpha.data_ptr(), w.data_ptr(), dsum.data_ptr(), dq.data_ptr(), ld(dq), du.data_ptr(), ld(du),
        None if dxr is None else dxr.data_ptr(), 0 if dxr is None else ld(dxr), n_dst, c, h, stream)
    _lib.check(st, "anemoi_gt_edge_attention_folded_backward_dst")
    rowptr_t, eid_t, dst_t, dst_of_edge = _transposed_csr(plan)
    st = lib.anemoi_gt_edge_attention_folded_backward_src(
        code, q.data_ptr(), ld(q), dout.data_ptr(), ld(dout), alpha.data_ptr(), w.data_ptr(), dsum.data_ptr(),
        rowptr_t.data_ptr(), eid_t.data_ptr(), dst_t.data_ptr(), dk.data_ptr(), dv.data_ptr(), ld(dk), n_src, c, h, stream)
    _lib.check(st, "anemoi_gt_edge_attention_folded_backward_src")
    if not want_dattr:
        return None
    dattr = torch.empty((n_edges, up), dtype=torch.float32, device=dev)
    st = lib.anemoi_gt_edge_attr_grad(code, alpha.data_ptr(), w.data_ptr(), dsum.data_ptr(), u.data_ptr(), ld(u),
                                      dt.data_ptr(), ld(dt), dst_of_edge.data_ptr(), dattr.data_ptr(), n_edges, h, up,
                                      c // h, stream)
    _lib.check(st, "anemoi_gt_edge_attr_grad")
    return dattr


def _edge_forward_lists(plan, q: Tensor) -> dict:
    """The plan's run lists (uniform-degree-3 decoder graphs) or destination schedule (mesh graphs) for the forward edge
    kernel -- what the inference route hands it (``layers/block.py::folded_edge_phase``); results equal the plain kernel's."""
    from .layers.block import edge_runs, edge_schedule

    runs = edge_runs(plan, q.dtype)
    return {"runs": runs, "sched": None if runs is not None else edge_schedule(plan, q)}  # (the tile kernel: inference route)


class _GTEdgeAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, x_r, u, edge_attr, plan, num_heads: int, up: int):
        lse = torch.empty((q.shape[0], num_heads), dtype=torch.float32, device=q.device)
        out = ops.gt_edge_attention_folded(q, k, v, x_r, u, edge_attr, plan.rowptr, plan.col, num_heads, up, lse=lse,
                                           **_edge_forward_lists(plan, q))
        ctx.save_for_backward(q, k, v, u, edge_attr, lse)  # the backward needs neither the result nor x_r
        ctx.plan, ctx.h, ctx.up, ctx.has_xr = plan, num_heads, up, x_r is not None
        return out

    @staticmethod
    def backward(ctx, dfull):
        q, k, v, u, edge_attr, lse = ctx.saved_tensors
        plan, h, up, has_xr = ctx.plan, ctx.h, ctx.up, ctx.has_xr
        n_dst, c = q.shape
        dfull = dfull.contiguous()
        dout = dfull[:, :c]
        dxr = dout.contiguous() if has_xr else None
        if plan.col.shape[0] == 0:  # no edges: out = x_r, t = 0 -- only x_r receives a gradient
            return (torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v), dxr, torch.zeros_like(u),
                    torch.zeros_like(edge_attr), None, None, None)
        dq, du = torch.empty_like(q), torch.empty((n_dst, h * up), dtype=q.dtype, device=q.device)
        dkv = torch.empty((k.shape[0], 2 * c), dtype=q.dtype, device=q.device)
        dattr = _edge_backward(q, k, v, dout, u, dfull[:, c:c + h * up], lse, edge_attr, plan, h, up, dq, dkv[:, :c],
                               dkv[:, c:], du, ctx.needs_input_grad[5])
        return dq, dkv[:, :c], dkv[:, c:], dxr, du, dattr, None, None, None


class _GTEdgeAttentionSelf(torch.autograd.Function):
    """The processor block's case: q, k, v, x_r, u are column ranges of ONE GEMM result ``sq = x_r | q | k | v | u``.  The
    backward kernels write their gradients straight into the column ranges of one ``d sq`` buffer (strided outputs), so
    autograd neither zero-fills nor accumulates five sliced gradients."""

    @staticmethod
    def forward(ctx, sq, edge_attr, plan, num_heads: int, up: int):
        c = (sq.shape[1] - num_heads * up) // 4
        lse = torch.empty((sq.shape[0], num_heads), dtype=torch.float32, device=sq.device)
        out = ops.gt_edge_attention_folded(sq[:, c:2 * c], sq[:, 2 * c:3 * c], sq[:, 3 * c:4 * c], sq[:, :c], sq[:, 4 * c:],
                                           edge_attr, plan.rowptr, plan.col, num_heads, up, lse=lse,
                                           **_edge_forward_lists(plan, sq[:, :c]))
        ctx.save_for_backward(sq, edge_attr, lse)
        ctx.plan, ctx.h, ctx.up, ctx.c = plan, num_heads, up, c
        return out

    @staticmethod
    def backward(ctx, dfull):
        sq, edge_attr, lse = ctx.saved_tensors
        plan, h, up, c = ctx.plan, ctx.h, ctx.up, ctx.c
        dfull = dfull.contiguous()
        dout = dfull[:, :c]
        dsq = torch.empty_like(sq)
        if plan.col.shape[0] == 0:
            dsq[:, :c].copy_(dout)  # d x_r
            dsq[:, c:].zero_()
            return dsq, torch.zeros_like(edge_attr), None, None, None
        dattr = _edge_backward(sq[:, c:2 * c], sq[:, 2 * c:3 * c], sq[:, 3 * c:4 * c], dout, sq[:, 4 * c:],
                               dfull[:, c:c + h * up], lse, edge_attr, plan, h, up, dsq[:, c:2 * c], dsq[:, 2 * c:3 * c],
                               dsq[:, 3 * c:4 * c], dsq[:, 4 * c:], ctx.needs_input_grad[1], dxr=dsq[:, :c])  # d x_r = dout
        return dsq, dattr, None, None, None


class _GTEdgeAttentionMapper(torch.autograd.Function):
    """The mapper block's case: ``sq = x_r | q | u`` ``[n_dst, 2C + H*up]`` and ``kv = k | v`` ``[n_src, 2C]`` are GEMM
    results consumed whole; the backward fills ONE ``d sq`` and ONE ``d kv`` buffer through strided outputs.  With five
    sliced inputs autograd zero-fills and adds a full-width buffer per slice -- on the 542 080 grid rows of config 3 that
    was 6 ms of fills and adds per training step."""

    @staticmethod
    def forward(ctx, sq, kv, edge_attr, plan, num_heads: int, up: int):
        c = kv.shape[1] // 2
        lse = torch.empty((sq.shape[0], num_heads), dtype=torch.float32, device=sq.device)
        out = ops.gt_edge_attention_folded(sq[:, c:2 * c], kv[:, :c], kv[:, c:], sq[:, :c], sq[:, 2 * c:], edge_attr,
                                           plan.rowptr, plan.col, num_heads, up, lse=lse, **_edge_forward_lists(plan, sq[:, :c]))
        ctx.save_for_backward(sq, kv, edge_attr, lse)
        ctx.plan, ctx.h, ctx.up, ctx.c = plan, num_heads, up, c
        return out

    @staticmethod
    def backward(ctx, dfull):
        sq, kv, edge_attr, lse = ctx.saved_tensors
        plan, h, up, c = ctx.plan, ctx.h, ctx.up, ctx.c
        dfull = dfull.contiguous()
        dout = dfull[:, :c]
        dsq = torch.empty_like(sq)
        if plan.col.shape[0] == 0:
            dsq[:, :c].copy_(dout)  # d x_r
            dsq[:, c:].zero_()
            return dsq, torch.zeros_like(kv), torch.zeros_like(edge_attr), None, None, None
        dkv = torch.empty_like(kv)
        dattr = _edge_backward(sq[:, c:2 * c], kv[:, :c], kv[:, c:], dout, sq[:, 2 * c:], dfull[:, c:c + h * up], lse,
                               edge_attr, plan, h, up, dsq[:, c:2 * c], dkv[:, :c], dkv[:, c:], dsq[:, 2 * c:],
                               ctx.needs_input_grad[2], dxr=dsq[:, :c])  # d x_r = dout
        return dsq, dkv, dattr, None, None, None


class _GTConv(torch.autograd.Function):
    """``GraphTransformerConv`` on explicit per-edge features ``e [E, C]`` (CSR order) ``+ x_r``: the route of the blocks
    for edge_dim values the folded kernels do not take (any edge_dim; reference layers/conv.py:98-142).  ``lin_edge`` is a
    differentiable Linear in front of it; backward = the two kernels of ``csrc/edge_backward.hip`` (explicit-edge
    variants), which also return ``d e``."""

    @staticmethod
    def forward(ctx, q, k, v, e, x_r, plan, num_heads: int, dropout_p: float = 0.0, seed: int = 0, seed_dev=None):
        lse = torch.empty((q.shape[0], num_heads), dtype=torch.float32, device=q.device)
        out = ops.gt_conv(q, k, v, e, plan.rowptr, plan.col, num_heads, x_r=x_r, lse=lse, dropout_p=dropout_p,
                          dropout_seed=seed, seed_dev=seed_dev)
        ctx.save_for_backward(q, k, v, e, lse)
        ctx.plan, ctx.h, ctx.has_xr = plan, num_heads, x_r is not None
        # (seed_dev is THIS call's word, see _MHSA: the backward rebuilds the mask its forward drew)
        ctx.drop = (float(dropout_p), int(seed) & 0xFFFFFFFF, seed_dev)
        return out

    @staticmethod
    def backward(ctx, dout):
        from . import _lib

        q, k, v, e, lse = ctx.saved_tensors
        plan, h = ctx.plan, ctx.h
        dout = dout.contiguous()
        dxr = dout if ctx.has_xr else None
        n_dst, c = q.shape
        n_src, n_edges = k.shape[0], plan.col.shape[0]
        if n_edges == 0:
            return (torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v), torch.zeros_like(e), dxr, None, None, None,
                    None, None)
        dev = q.device
        p_drop, seed, seed_dev = ctx.drop
        drop = (p_drop, seed, ops._seed_dev_ptr(seed_dev, q))
        alpha = torch.empty((n_edges, h), dtype=torch.float32, device=dev)
        w = torch.empty((n_edges, h), dtype=torch.float32, device=dev)
        dsum = torch.empty((n_dst, h), dtype=torch.float32, device=dev)
        dq = torch.empty((n_dst, c), dtype=q.dtype, device=dev)
        dkv = torch.empty((n_src, 2 * c), dtype=q.dtype, device=dev)
        de = torch.empty((n_edges, c), dtype=q.dtype, device=dev)
        lib, code, stream = _lib.load(), ops.dtype_code(q.dtype), ops._stream()
        ld = lambda t: ops._ld(ops._rows(t))  # noqa: E731
        if ld(k) != ld(v):
            raise ValueError("gt_conv: k and v must share their leading dimension (slices of one k | v buffer)")
        st = lib.anemoi_gt_conv_backward_dst(code, q.data_ptr(), ld(q), k.data_ptr(), v.data_ptr(), ld(k), e.data_ptr(), ld(e),
                                             dout.data_ptr(), ld(dout), lse.data_ptr(), plan.rowptr.data_ptr(),
                                             plan.col.data_ptr(), alpha.data_ptr(), w.data_ptr(), dsum.data_ptr(),
                                             dq.data_ptr(), c, n_dst, c, h, *drop, stream)
        _lib.check(st, "anemoi_gt_conv_backward_dst")
        rowptr_t, eid_t, dst_t, _ = _transposed_csr(plan)
        st = lib.anemoi_gt_conv_backward_src(code, q.data_ptr(), ld(q), dout.data_ptr(), ld(dout), alpha.data_ptr(),
                                             w.data_ptr(), dsum.data_ptr(), rowptr_t.data_ptr(), eid_t.data_ptr(),
                                             dst_t.data_ptr(), dkv.data_ptr(), dkv[:, c:].data_ptr(), 2 * c, de.data_ptr(), c,
                                             n_src, c, h, *drop, stream)
        _lib.check(st, "anemoi_gt_conv_backward_src")
        return dq, dkv[:, :c], dkv[:, c:], de, dxr, None, None, None, None, None


def gt_conv(q: Tensor, k: Tensor, v: Tensor, e_csr: Tensor, x_r: Optional[Tensor], plan, num_heads: int,
            dropout_p: float = 0.0, seed: Optional[int] = None, seed_dev: Optional[Tensor] = None) -> Tensor:
    """Differentiable ``ops.gt_conv``: gradients for ``q, k, v`` (``k`` / ``v`` slices of one buffer), the per-edge
    features ``e_csr [E, C]`` and ``x_r``.  ``dropout_p > 0``: dropout of the attention weights (reference layers/conv.py:140)
    with the mask derived from ``seed`` (default: drawn from torch's CPU generator, so ``torch.manual_seed`` reproduces it);
    the backward rebuilds the same mask."""
    if dropout_p > 0.0 and seed is None:
        seed = int(torch.randint(0, 2**31 - 1, (1,)).item())
    drop = (float(dropout_p), int(seed or 0), seed_dev)
    out_dtype, c = q.dtype, q.shape[1]
    if _edge_phase_in_f32(q.dtype, c, num_heads):
        kv = torch.cat([k, v], dim=1).float()
        q, k, v, e_csr, x_r = q.float(), kv[:, :c], kv[:, c:], e_csr.float(), None if x_r is None else x_r.float()
    d = c // num_heads
    d_pad = conv_head_size(d, q.dtype)
    if d_pad != d:
        # A head size the kernels' lane groups do not come in (they reduce over 1, 2, 4, 8 or 16 lanes of 16 bytes: D = 12, or
        # D = 5 as the reference's tests like them): every head zero-padded to the next size that does.  Zero columns add
        # nothing to q . (k + e) and produce zero output columns (dropped below); the kernels scale by 1 / sqrt(their D), so q
        # carries sqrt(D_pad / D).  Plain torch ops around the same autograd node: the gradients of the real columns are
        # exact, those of the padding are discarded by the slice.
        def pad(t):
            return torch.nn.functional.pad(t.reshape(t.shape[0], num_heads, d), (0, d_pad - d)).reshape(t.shape[0], num_heads * d_pad)

        kvp = torch.cat([pad(k), pad(v)], dim=1)
        cp = num_heads * d_pad
        out = _GTConv.apply(pad(q) * (d_pad / d) ** 0.5, kvp[:, :cp], kvp[:, cp:], pad(e_csr),
                            None if x_r is None else pad(x_r), plan, num_heads, *drop)
        out = out.reshape(out.shape[0], num_heads, d_pad)[:, :, :d].reshape(out.shape[0], c).contiguous()  # (D = 1: a strided view)
    else:
        out = _GTConv.apply(q, k, v, e_csr, x_r, plan, num_heads, *drop)
    return out if out.dtype == out_dtype else out.to(out_dtype)


def conv_head_size(d: int, dtype: torch.dtype) -> int:
    """The head size the explicit-edge conv kernels run ``d`` at: lanes of 16 bytes (4 f32 / 8 bf16 channels) in groups of
    1, 2, 4, 8 or 16 per head -- ``d`` itself where it is such a size."""
    vec = 16 // torch.empty((), dtype=dtype).element_size()
    lanes = max(1, -(-d // vec))
    lanes = 1 << (lanes - 1).bit_length()
    if lanes > 16:
        raise NotImplementedError(f"GraphTransformerConv: head size {d} beyond {16 * vec} channels per head")
    return vec * lanes


def _edge_phase_in_f32(dtype: torch.dtype, c: int, num_heads: int) -> bool:
    """bf16 head sizes that are a multiple of 4 but not of 8 (BASELINE config 1: 64 channels / 16 heads): the bf16 edge
    kernels move 8 channels per lane, so the edge phase of such a block runs on the f32 kernels (4 channels per lane) between
    two casts -- the GEMMs around it stay bf16.  Tiny heads mean a tiny model: the casts cost nothing that matters."""
    d = c // num_heads
    return dtype == torch.bfloat16 and d % 8 != 0 and d % 4 == 0


def gt_edge_attention_packed(sq: Tensor, kv: Tensor, edge_attr: Tensor, plan, num_heads: int, up: int) -> Tensor:
    """:func:`gt_edge_attention` on the packed GEMM results ``sq = x_r | q | u`` ``[n_dst, 2C + H*up]`` and
    ``kv = k | v`` ``[n_src, 2C]`` (the mapper blocks' layout): gradients arrive as one ``d sq`` and one ``d kv``."""
    if _edge_phase_in_f32(sq.dtype, kv.shape[1] // 2, num_heads):
        return _GTEdgeAttentionMapper.apply(sq.float(), kv.float(), edge_attr, plan, num_heads, up).to(sq.dtype)
    return _GTEdgeAttentionMapper.apply(sq, kv, edge_attr, plan, num_heads, up)


def gt_edge_attention(q: Tensor, k: Tensor, v: Tensor, x_r: Optional[Tensor], u: Tensor, edge_attr: Tensor, plan,
                      num_heads: int, up: int) -> Tensor:
    """Differentiable ``ops.gt_edge_attention_folded``: ``[n_dst, C + H*up] = [sum alpha v (+ x_r) | sum alpha a]`` with
    gradients for ``q, k, v, x_r, u`` (compute dtype) and ``edge_attr`` (f32 ``[E, up]``, CSR order of ``plan``)."""
    if _edge_phase_in_f32(q.dtype, q.shape[1], num_heads):
        kv = torch.cat([k, v], dim=1).float()  # (the kernels want k and v as column ranges of one buffer)
        c = q.shape[1]
        return _GTEdgeAttention.apply(q.float(), kv[:, :c], kv[:, c:], None if x_r is None else x_r.float(), u.float(),
                                      edge_attr, plan, num_heads, up).to(q.dtype)
    return _GTEdgeAttention.apply(q, k, v, x_r, u, edge_attr, plan, num_heads, up)


# ------------------------------------------------------------------------------------------ mesh-node self attention
class _MHSA(torch.autograd.Function):
    """``dropout(softmax(Q K^T / sqrt(D))) V`` on the fused ``q | k | v`` matrix (``anemoi_mhsa``, MFMA flash kernels for
    bf16 head sizes 64 / 32, with or without dropout); the backward recomputes the probabilities from the saved
    log-sum-exp and the dropout mask from the saved seed (``anemoi_mhsa_backward``)."""

    @staticmethod
    def forward(ctx, qkv: Tensor, batch_size: int, num_heads: int, window: int, dropout_p: float, seed: int,
                head_offset: int = 0, heads_total: int = 0, seed_dev: Optional[Tensor] = None):
        out, lse = ops.mhsa(qkv, batch_size, num_heads, window, return_lse=True, dropout_p=dropout_p, dropout_seed=seed,
                            head_offset=head_offset, heads_total=heads_total, seed_dev=seed_dev)
        ctx.save_for_backward(qkv, out, lse)
        # (seed_dev is THIS call's word -- runtime.DeviceDropout.seed_word() makes one per call --, so the backward reads
        #  the value the forward saw whatever the step counter has done since)
        ctx.args = (batch_size, num_heads, window, dropout_p, seed, head_offset, heads_total, seed_dev)
        return out

    @staticmethod
    def backward(ctx, dout: Tensor):
        qkv, out, lse = ctx.saved_tensors
        b, h, w, p, seed, h0, ht, seed_dev = ctx.args
        return (ops.mhsa_backward(qkv, out, dout.contiguous(), lse, b, h, w, p, seed, h0, ht, seed_dev=seed_dev), None, None,
                None, None, None, None, None, None)


def mhsa(qkv: Tensor, batch_size: int, num_heads: int, window: int = -1, dropout_p: float = 0.0,
         seed: Optional[int] = None, head_offset: int = 0, heads_total: int = 0, seed_dev: Optional[Tensor] = None) -> Tensor:
    """Differentiable ``ops.mhsa`` (reference layers/attention.py:67-112).  ``dropout_p`` > 0: attention dropout with a
    mask derived from ``seed`` (default: drawn from torch's CPU generator, so ``torch.manual_seed`` reproduces it);
    ``head_offset`` / ``heads_total``: a head shard of a model group (see ``ops.mhsa``)."""
    if dropout_p > 0.0 and seed is None:
        seed = int(torch.randint(0, 2**31 - 1, (1,)).item())
    return _MHSA.apply(qkv, batch_size, num_heads, window, float(dropout_p), int(seed or 0), int(head_offset),
                       int(heads_total), seed_dev)


# ------------------------------------------------------------------------------------------ GNN edge phase
class _GatherAddAct(torch.autograd.Function):
    """``act(t[e] + p_dst[dst[e]] + p_src[src[e]])`` over the CSR slots of ``plan`` (``anemoi_gather_add_act``).  Backward:
    the pre-activation is recomputed by the same kernel, ``d t = d out * act'(pre)``; ``d p_dst`` is the segment sum of
    ``d t`` over the destination-sorted CSR, ``d p_src`` the segment sum over the source-major (transposed) CSR -- no
    atomics, reproducible bit for bit."""

    @staticmethod
    def forward(ctx, t, p_dst, p_src, plan, act: str):
        out = ops.gather_add_act(t, p_dst, p_src, plan.dst, plan.col, act=act)
        ctx.save_for_backward(t, p_dst, p_src)
        ctx.plan, ctx.act = plan, act
        return out

    @staticmethod
    def backward(ctx, dout):
        t, p_dst, p_src = ctx.saved_tensors
        plan, act = ctx.plan, ctx.act
        dout = dout.contiguous()
        if act == "Identity":
            dpre = dout
        else:
            pre = ops.gather_add_act(t, p_dst, p_src, plan.dst, plan.col, act="Identity")
            if pre.shape[1] % (16 // pre.element_size()) == 0:
                dpre = ops.act_backward(pre, dout, act)
            else:  # narrow rows: torch derives the activation (plumbing-sized tensors only)
                with torch.enable_grad():
                    pr = pre.float().requires_grad_()
                    (dpre,) = torch.autograd.grad(_TORCH_ACT[act](pr), pr, dout.float())
                dpre = dpre.to(dout.dtype)
        d_dst = d_src = None
        if ctx.needs_input_grad[1]:
            d_dst = ops.segment_sum(dpre, plan.rowptr)
        if ctx.needs_input_grad[2]:
            rowptr_t, eid_t, _, _ = _transposed_csr(plan)
            d_src = ops.segment_sum(dpre.index_select(0, eid_t.long()), rowptr_t)
        return dpre, d_dst, d_src, None, None


class _SegmentSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, plan):
        ctx.plan = plan
        return ops.segment_sum(v, plan.rowptr)

    @staticmethod
    def backward(ctx, dout):
        return dout.index_select(0, ctx.plan.dst.long()), None


def gather_add_act(t: Tensor, p_dst: Tensor, p_src: Tensor, plan, act: str = "Identity") -> Tensor:
    """Differentiable ``ops.gather_add_act`` (first edge-MLP layer of the GNN blocks, reference layers/conv.py:47-76)."""
    return _GatherAddAct.apply(t, p_dst, p_src, plan, act)


def segment_sum(v: Tensor, plan) -> Tensor:
    """Differentiable ``ops.segment_sum``: the scatter-sum of edge rows over their destinations (reference
    layers/conv.py:74, PyG ``scatter(reduce="sum")``)."""
    return _SegmentSum.apply(v, plan)


# ------------------------------------------------------------------------------------------ a whole processor block
def _lin_edge_fold(sd: dict, prefix: str, c: int, h: int, up: int, device):
    """(W_u [H*up, C], b_u [H*up], W_t [C, H*up]) of the lin_edge fold, differentiable w.r.t. lin_edge / lin_query /
    projection (see layers/block.py::_query_fold, _projection_fold of this package)."""
    g = lambda name: sd[prefix + "." + name]  # noqa: E731
    d = c // h
    edge_dim = g("lin_edge.weight").shape[1]
    pad = torch.zeros((c, up - edge_dim - 1), dtype=torch.float32, device=device)
    weh = torch.cat([g("lin_edge.weight"), g("lin_edge.bias")[:, None], pad], dim=1).view(h, d, up)
    w_u = torch.einsum("hda,hdc->hac", weh, g("lin_query.weight").view(h, d, c)).reshape(h * up, c)
    b_u = torch.einsum("hda,hd->ha", weh, g("lin_query.bias").view(h, d)).reshape(h * up)
    w_t = torch.einsum("ohd,hda->oha", g("projection.weight").view(c, h, d), weh).reshape(c, h * up)
    return w_u, b_u, w_t


FOLD_MAX_UP = 16  # widest per-head edge representation of the folded edge kernels (edge_dim + 1 rounded up to 4)


def folded_edge_route(dtype: torch.dtype, c: int, num_heads: int, up: int) -> bool:
    """Do the differentiable blocks take the FOLDED edge kernels at this shape?  They exist for a per-head edge width up to
    ``FOLD_MAX_UP``, head sizes of 1, 2, 4, 8 or 16 lanes of 16 bytes (in the dtype the edge phase runs in: bf16 heads of 4
    run it in f32) and a packed ``u`` / ``t`` width ``H * up`` that keeps the rows 16-byte aligned (one bf16 head with 3 or 11
    edge attributes does not).  Everything else -- many edge attributes, head sizes like 12 / 20 / 48 / 96, a single bf16
    head -- runs ``lin_edge`` as a GEMM and the conv on explicit per-edge features (``gt_conv``: any head size, zero-padded).
    Round 6: until then only ``up`` decided, and the other shapes raised from the kernel's dispatch in training mode."""
    if up > FOLD_MAX_UP or c % num_heads != 0:
        return False
    edge_dtype = torch.float32 if _edge_phase_in_f32(dtype, c, num_heads) else dtype
    vec = 16 // torch.empty((), dtype=edge_dtype).element_size()
    d = c // num_heads
    return d % vec == 0 and (d // vec) in (1, 2, 4, 8, 16) and (num_heads * up) % vec == 0


class GTBlockWeights:
    """The operands of ONE processor block out of :func:`gt_processor_weights`: f32 ``w_in`` / ``b_in`` / ``w_p`` (results
    of the batched fold algebra -- autograd carries their gradients back to every block's parameters in one batched
    backward) and the :class:`WeightPrep` of the block's four GEMMs."""

    __slots__ = ("w_in", "b_in", "w_p", "p_in", "p_p", "p_1", "p_2")


def _stacked_cast_and_transpose(stacked: Tensor, dtype: torch.dtype):
    """``[L, N, K]`` f32 -> (``[L, N, K]``, ``[L, K, N]``) in the compute dtype: one cast + one chunked transpose launch."""
    from . import _lib

    count, n, k = stacked.shape
    w = stacked.detach().to(dtype)
    wt = torch.empty((count, k, n), dtype=dtype, device=w.device)
    st = _lib.load().anemoi_transpose_chunked(ops.dtype_code(dtype), w.data_ptr(), k, wt.data_ptr(), n, count * n, k, n, None,
                                              ops._stream())
    _lib.check(st, "anemoi_transpose_chunked")
    return w, wt


def _params_cast_and_transpose(params, dtype: torch.dtype):
    """The same for L separate ``[N, K]`` f32 parameters: a multi-tensor cast into one buffer + one chunked transpose."""
    from . import _lib

    count = len(params)
    n, k = params[0].shape
    w = torch.empty((count, n, k), dtype=dtype, device=params[0].device)
    torch._foreach_copy_(list(w.unbind(0)), [p.detach() for p in params])
    wt = torch.empty((count, k, n), dtype=dtype, device=w.device)
    st = _lib.load().anemoi_transpose_chunked(ops.dtype_code(dtype), w.data_ptr(), k, wt.data_ptr(), n, count * n, k, n, None,
                                              ops._stream())
    _lib.check(st, "anemoi_transpose_chunked")
    return w, wt


def _gt_stacked_fold(sds: list, prefix: str, c: int, h: int, up: int, device):
    """``(w_in [L, 4C + H*up, C], b_in [L, 4C + H*up], w_p [L, C, C + H*up])`` in f32 for the L blocks ``sds``: per block the
    rows ``lin_self | lin_query | lin_key | lin_value | W_u`` (+ biases, ``b_u`` last) and the columns ``projection | W_t`` with
    ``(W_u, b_u, W_t)`` the lin_edge fold of :func:`_lin_edge_fold` -- the same values, from three batched einsums and three
    concatenations for all blocks (plain torch on the parameters: autograd carries the gradients back)."""
    count = len(sds)
    key = lambda sd, n, part: sd[f"{prefix}.{n}.{part}"]  # noqa: E731
    col = lambda n, part: [key(sd, n, part) for sd in sds]  # noqa: E731
    d = c // h
    we, be = torch.stack(col("lin_edge", "weight")), torch.stack(col("lin_edge", "bias"))
    edge_dim = we.shape[2]
    pad = torch.zeros((count, c, up - edge_dim - 1), dtype=torch.float32, device=device)
    weh = torch.cat([we, be[:, :, None], pad], dim=2).view(count, h, d, up)
    wq, bq = torch.stack(col("lin_query", "weight")), torch.stack(col("lin_query", "bias"))
    wp = torch.stack(col("projection", "weight"))
    w_u = torch.einsum("lhda,lhdc->lhac", weh, wq.view(count, h, d, c)).reshape(count, h * up, c)
    b_u = torch.einsum("lhda,lhd->lha", weh, bq.view(count, h, d)).reshape(count, h * up)
    w_t = torch.einsum("lohd,lhda->loha", wp.view(count, c, h, d), weh).reshape(count, c, h * up)
    n_in = 4 * c + h * up
    rows, vecs = [], []
    w_u_i, b_u_i = w_u.unbind(0), b_u.unbind(0)  # (unbind: ONE backward node that stacks the L gradients)
    for i, sd in enumerate(sds):  # one cat over 5 L matrices: block i's rows are x_r | q | k | v | u
        rows += [key(sd, "lin_self", "weight"), key(sd, "lin_query", "weight"), key(sd, "lin_key", "weight"),
                 key(sd, "lin_value", "weight"), w_u_i[i]]
        vecs += [key(sd, "lin_self", "bias"), key(sd, "lin_query", "bias"), key(sd, "lin_key", "bias"),
                 key(sd, "lin_value", "bias"), b_u_i[i]]
    w_in = torch.cat(rows, dim=0).view(count, n_in, c)
    b_in = torch.cat(vecs, dim=0).view(count, n_in)
    w_p = torch.cat([wp, w_t], dim=2)  # [L, C, C + H * up]
    return w_in, b_in, w_p


def gt_processor_weights(sds: list, prefix: str, c: int, h: int, up: int, dtype: torch.dtype, device):
    """Everything the blocks of a GraphTransformer processor derive from their PARAMETERS, for all blocks at once: the
    lin_edge fold (three batched einsums instead of three per block), ``x_r|q|k|v|u`` / ``projection|t`` weight assembly,
    the casts to the compute dtype and the transposes the dX GEMMs read.  Per block and step the per-block route spends
    ~50 launches of a few microseconds on this (a third of the launches of a config-3 training step); here it is ~40
    launches per processor.  Returns a list of :class:`GTBlockWeights`, or None when the blocks do not qualify (other dtype
    than bf16, widths off the GEMM's 64-element slab, differing shapes) -- the caller then takes the per-block route.
    ``ANEMOI_AMD_TRAIN_BATCHED_PARAMS=0`` switches it off (A/B runs)."""
    import os

    count = len(sds)
    km = ops.k_multiple(dtype)
    if (os.environ.get("ANEMOI_AMD_TRAIN_BATCHED_PARAMS", "1") == "0" or dtype != torch.bfloat16 or count < 2
            or not folded_edge_route(dtype, c, h, up) or c % km != 0 or (h * up) % km != 0):
        return None
    names = ("lin_self", "lin_query", "lin_key", "lin_value", "lin_edge", "projection", "node_dst_mlp.1", "node_dst_mlp.3")
    key = lambda sd, n, part: sd.get(f"{prefix}.{n}.{part}")  # noqa: E731
    for n in names:
        ws, bs = [key(sd, n, "weight") for sd in sds], [key(sd, n, "bias") for sd in sds]
        if any(w is None or b is None or w.dtype != torch.float32 or b.dtype != torch.float32 or w.shape != ws[0].shape
               or not w.is_contiguous() for w, b in zip(ws, bs)):
            return None
    hidden = key(sds[0], "node_dst_mlp.1", "weight").shape[0]
    if hidden % km != 0 or key(sds[0], "node_dst_mlp.3", "weight").shape != (c, hidden):
        return None
    col = lambda n, part: [key(sd, n, part) for sd in sds]  # noqa: E731
    w_in, b_in, w_p = _gt_stacked_fold(sds, prefix, c, h, up, device)
    n_in = 4 * c + h * up
    sink_in = GradSink(count, n_in, c, True, device, stacked_parts=2)
    sink_p = GradSink(count, c, c + h * up, True, device, stacked_parts=1)  # (the projection's bias is a leaf: its slot goes to it)
    w_in_c, w_in_t = _stacked_cast_and_transpose(w_in, dtype)
    w_p_c, w_p_t = _stacked_cast_and_transpose(w_p, dtype)
    w1_c, w1_t = _params_cast_and_transpose(col("node_dst_mlp.1", "weight"), dtype)
    w2_c, w2_t = _params_cast_and_transpose(col("node_dst_mlp.3", "weight"), dtype)
    w_in_i = _Unstack.apply(w_in, sink_in, "w")
    b_in_i = _Unstack.apply(b_in, sink_in, "b")
    w_p_i = _Unstack.apply(w_p, sink_p, "w")
    out = []
    for i in range(count):
        bw = GTBlockWeights()
        bw.w_in, bw.b_in, bw.w_p = w_in_i[i], b_in_i[i], w_p_i[i]
        bw.p_in = WeightPrep(w_in_c[i], w_in_t[i], sink_in, i)
        bw.p_p = WeightPrep(w_p_c[i], w_p_t[i], sink_p, i)
        bw.p_1 = WeightPrep(w1_c[i], w1_t[i])
        bw.p_2 = WeightPrep(w2_c[i], w2_t[i])
        out.append(bw)
    return out


def _explicit_edge_features(sd: dict, prefix: str, edge_attr_csr: Tensor, dtype: torch.dtype) -> Tensor:
    """``lin_edge(a)`` as an explicit ``[E, C]`` matrix in CSR order (the route for ``up > FOLD_MAX_UP``): the attribute
    matrix carries a constant 1 behind the ``edge_dim`` real columns, so ``[W_e | b_e | 0]`` is the whole Linear."""
    w_e, b_e = sd[prefix + ".lin_edge.weight"], sd[prefix + ".lin_edge.bias"]
    up, edge_dim = edge_attr_csr.shape[1], w_e.shape[1]
    pad = torch.zeros((w_e.shape[0], up - edge_dim - 1), dtype=w_e.dtype, device=w_e.device)
    return linear(edge_attr_csr.to(dtype), torch.cat([w_e, b_e[:, None], pad], dim=1), None)


def _gt_tail(y_att: Tensor, x_skip: Tensor, sd: dict, prefix: str, w_t: Optional[Tensor], act: str, eps: float,
             prepared: Optional[GTBlockWeights] = None) -> Tensor:
    g = lambda name: sd[prefix + "." + name]  # noqa: E731
    if prepared is not None:
        y = linear(y_att, prepared.w_p, g("projection.bias"), "Identity", x_skip, prep=prepared.p_p)
    else:
        w_p = g("projection.weight") if w_t is None else torch.cat([g("projection.weight"), w_t], dim=1)
        y = linear(y_att, w_p, g("projection.bias"), "Identity", x_skip)
    h1, y = layer_norm_skip(y, g("node_dst_mlp.0.weight"), g("node_dst_mlp.0.bias"), eps)
    return mlp2(h1, g("node_dst_mlp.1.weight"), g("node_dst_mlp.1.bias"), g("node_dst_mlp.3.weight"),
                g("node_dst_mlp.3.bias"), act, y, None if prepared is None else prepared.p_1,
                None if prepared is None else prepared.p_2)


def gt_processor_block(x: Tensor, sd: dict, prefix: str, edge_attr_csr: Tensor, plan, num_heads: int,
                       act: str = "GELU", eps: float = 1e-5, prepared: Optional[GTBlockWeights] = None) -> Tensor:
    """Differentiable ``GraphTransformerProcessorBlock`` (reference layers/block.py:602-635) on the HIP kernels:
    ``sd[prefix + ".lin_query.weight"]`` etc. are the block's f32 parameters (``requires_grad`` as wanted),
    ``edge_attr_csr`` ``[E, up]`` f32 the edge attributes in the plan's CSR order with the constant-1 column behind the
    ``edge_dim`` real ones (``ops.edge_attr_csr``).  ``lin_edge`` is folded into the q GEMM and the projection exactly
    as in the inference path; the fold itself is ordinary torch algebra on the parameters, so autograd carries the
    gradients back to ``lin_edge`` / ``lin_query`` / ``projection``.  ``prepared``: this block's entry of
    :func:`gt_processor_weights` (the fold algebra, casts and transposes done for all blocks of the processor at once)."""
    g = lambda name: sd[prefix + "." + name]  # noqa: E731
    c = x.shape[1]
    h, up = num_heads, edge_attr_csr.shape[1]
    if prepared is not None:
        xh, x = layer_norm_skip(x, g("layer_norm1.weight"), g("layer_norm1.bias"), eps)
        sq = linear(xh, prepared.w_in, prepared.b_in, prep=prepared.p_in)  # x_r | q | k | v | u
        if _edge_phase_in_f32(sq.dtype, (sq.shape[1] - h * up) // 4, h):
            att = _GTEdgeAttentionSelf.apply(sq.float(), edge_attr_csr, plan, h, up).to(sq.dtype)
        else:
            att = _GTEdgeAttentionSelf.apply(sq, edge_attr_csr, plan, h, up)
        return _gt_tail(att, x, sd, prefix, None, act, eps, prepared)
    if not folded_edge_route(x.dtype, c, h, up):  # lin_edge as a GEMM, the conv on explicit per-edge features
        xh = layer_norm(x, g("layer_norm1.weight"), g("layer_norm1.bias"), eps)
        sq = linear(xh, torch.cat([g("lin_self.weight"), g("lin_query.weight"), g("lin_key.weight"), g("lin_value.weight")], 0),
                    torch.cat([g("lin_self.bias"), g("lin_query.bias"), g("lin_key.bias"), g("lin_value.bias")], 0))
        if plan.col.shape[0] == 0:  # an edge set without edges: the conv's sums are empty, x_r passes through
            att = sq[:, :c].contiguous()
        else:
            att = gt_conv(sq[:, c:2 * c], sq[:, 2 * c:3 * c], sq[:, 3 * c:],
                          _explicit_edge_features(sd, prefix, edge_attr_csr, x.dtype), sq[:, :c], plan, h)
        return _gt_tail(att, x, sd, prefix, None, act, eps)
    w_u, b_u, w_t = _lin_edge_fold(sd, prefix, c, h, up, x.device)
    w_in = torch.cat([g("lin_self.weight"), g("lin_query.weight"), g("lin_key.weight"), g("lin_value.weight"), w_u], 0)
    b_in = torch.cat([g("lin_self.bias"), g("lin_query.bias"), g("lin_key.bias"), g("lin_value.bias"), b_u], 0)
    xh, x = layer_norm_skip(x, g("layer_norm1.weight"), g("layer_norm1.bias"), eps)
    sq = linear(xh, w_in, b_in)  # x_r | q | k | v | u
    if _edge_phase_in_f32(sq.dtype, (sq.shape[1] - h * up) // 4, h):
        att = _GTEdgeAttentionSelf.apply(sq.float(), edge_attr_csr, plan, h, up).to(sq.dtype)
    else:
        att = _GTEdgeAttentionSelf.apply(sq, edge_attr_csr, plan, h, up)
    return _gt_tail(att, x, sd, prefix, w_t, act, eps)


def gt_mapper_block(x_src: Optional[Tensor], x_dst: Tensor, sd: dict, prefix: str, edge_attr_csr: Tensor, plan, num_heads: int,
                    act: str = "GELU", eps: float = 1e-5, kv_fn=None, sq_fn=None) -> Tensor:
    """Differentiable ``GraphTransformerMapperBlock`` (reference layers/block.py:479-550, ``update_src_nodes=False``):
    keys / values from ``LayerNorm1(x_src)``, queries / self term from ``LayerNorm2(x_dst)``, the new destination nodes
    are returned; same kernels and fold as :func:`gt_processor_block`."""
    g = lambda name: sd[prefix + "." + name]  # noqa: E731
    c = x_dst.shape[1]
    h, up = num_heads, edge_attr_csr.shape[1]
    # ``kv_fn`` / ``sq_fn`` (the mappers, training.gt_mapper): ``Linear(LayerNorm(.))`` of the source / destination rows
    # computed on the RAW node features (folded_embedding_ln_linear) -- called with (weight rows, bias); x_src may then
    # be None (its embedding is never formed)
    w_kv = torch.cat([g("lin_key.weight"), g("lin_value.weight")], 0)
    b_kv = torch.cat([g("lin_key.bias"), g("lin_value.bias")], 0)
    if kv_fn is not None:
        kv = kv_fn(w_kv, b_kv, g("layer_norm1.weight"), g("layer_norm1.bias"))
    else:
        kv = linear(layer_norm(x_src, g("layer_norm1.weight"), g("layer_norm1.bias"), eps), w_kv, b_kv)
    folded = folded_edge_route(x_dst.dtype, c, h, up)
    if sq_fn is None or not folded:
        xd, x_dst = layer_norm_skip(x_dst, g("layer_norm2.weight"), g("layer_norm2.bias"), eps)
    if not folded:  # see gt_processor_block
        sq = linear(xd, torch.cat([g("lin_self.weight"), g("lin_query.weight")], 0),
                    torch.cat([g("lin_self.bias"), g("lin_query.bias")], 0))
        if plan.col.shape[0] == 0:  # (see gt_processor_block)
            att = sq[:, :c].contiguous()
        else:
            att = gt_conv(sq[:, c:], kv[:, :c], kv[:, c:], _explicit_edge_features(sd, prefix, edge_attr_csr, x_dst.dtype),
                          sq[:, :c], plan, h)
        return _gt_tail(att, x_dst, sd, prefix, None, act, eps)
    w_u, b_u, w_t = _lin_edge_fold(sd, prefix, c, h, up, x_dst.device)
    w_sq = torch.cat([g("lin_self.weight"), g("lin_query.weight"), w_u], 0)
    b_sq = torch.cat([g("lin_self.bias"), g("lin_query.bias"), b_u], 0)
    if sq_fn is not None:
        sq = sq_fn(w_sq, b_sq, g("layer_norm2.weight"), g("layer_norm2.bias"))  # x_r | q | u
    else:
        sq = linear(xd, w_sq, b_sq)  # x_r | q | u
    att = gt_edge_attention_packed(sq, kv, edge_attr_csr, plan, h, up)
    return _gt_tail(att, x_dst, sd, prefix, w_t, act, eps)


# ------------------------------------------------------------------------------------------ the whole flat model
class _PermuteRows(torch.autograd.Function):
    """``x[perm]`` for a PERMUTATION ``perm`` of the rows (an edge plan's ``perm`` / its inverse): the backward is the
    inverse row copy, not the sort-based accumulation torch derives for a general advanced index (1.7 ms per step on the
    542 080-edge attribute matrices of config 3)."""

    @staticmethod
    def forward(ctx, x: Tensor, perm: Tensor):
        if perm.shape[0] != x.shape[0]:
            raise ValueError("permute_rows: the index must be a permutation of the rows")
        ctx.save_for_backward(perm)
        return x.index_select(0, perm)

    @staticmethod
    def backward(ctx, g: Tensor):
        (perm,) = ctx.saved_tensors
        out = torch.empty_like(g)
        out.index_copy_(0, perm, g)
        return out, None


def permute_rows(x: Tensor, perm: Tensor) -> Tensor:
    """Differentiable ``x[perm]`` for a row permutation ``perm`` (int64)."""
    if x.shape[0] == 0 and perm.numel() == 0:  # (an edge set without edges)
        return x
    return _PermuteRows.apply(x, perm)


def _edge_attr_csr(edge_attr_buf: Tensor, trainable: Optional[Tensor], plan, up: int, batch_size: int = 1) -> Tensor:
    """``[edge_attr | trainable | 1 | 0-pad]`` f32 in the plan's CSR order (what ``anemoi_edge_attr_csr`` builds in the
    inference path), as torch ops: differentiable w.r.t. the trainable edge tensor.  ``batch_size`` > 1: the plan spans
    the batched graph, the attributes repeat per sample (reference layers/graph.py:37-44)."""
    parts = [edge_attr_buf.float()] + ([] if trainable is None else [trainable.float()])
    attr = torch.cat(parts, dim=1).repeat(batch_size, 1)
    perm = plan.perm.long()
    # a whole-graph plan permutes the rows; a rank-local plan of the node-partitioned run selects its edges
    attr = permute_rows(attr, perm) if perm.shape[0] == attr.shape[0] else attr[perm]
    e, dim = attr.shape
    tail = torch.zeros((e, up - dim), dtype=torch.float32, device=attr.device)
    tail[:, 0] = 1.0
    return torch.cat([attr, tail], dim=1).contiguous()


def model_forward(sd: dict, graph: dict, x: Tensor, *, num_heads: int, num_layers: int, num_chunks: int,
                  prognostic_in, prognostic_out, dtype: torch.dtype = torch.float32, act: str = "GELU",
                  data: str = "data", hidden: str = "hidden", plan_cache=None) -> Tensor:
    """Differentiable forward of the flat GraphTransformer ``AnemoiModelEncProcDec`` (reference
    models/encoder_processor_decoder.py:168-233; batch size 1, no boundings) for TRAINING on the HIP kernels.

    ``sd``: the model's ``state_dict`` as f32 tensors on the device (``requires_grad`` where gradients are wanted, the
    trainable node / edge tensors included); ``graph``: ``{enc,proc,dec}_edge_index`` (int64 ``[2, E]``) and
    ``{enc,proc,dec}_edge_attr`` (f32 ``[E, k]``) as in ``oracle.reference_path.model_forward``.  Heavy ops are the
    autograd Functions above; concatenations / index maps are torch glue.  ``plan_cache`` (a ``runtime.PlanCache``, the
    model passes its own) keeps the CSR plans -- and the transposed CSR the backward hangs on them -- across steps: without
    it every step pays the range-check host syncs and two sorts per edge set."""
    from . import runtime

    b, t, ens, g_, v = x.shape
    if ens != 1 and b != 1:
        raise NotImplementedError("autograd.model_forward: an ensemble dimension > 1 only with batch size 1 (the "
                                  "reference repeats the node attributes per batch element only)")
    bs = b * ens  # rows are ordered (batch, ensemble, grid) as in the reference's rearrange (:173-177)
    # (any head size: the blocks take the folded edge kernels where they exist for it, else the conv on explicit edge features
    #  with zero-padded heads -- folded_edge_route)

    def node_attrs(name):
        parts = [sd[f"node_attributes.latlons_{name}"]]
        tr = sd.get(f"node_attributes.trainable_tensors.{name}.trainable")
        return torch.cat(parts + ([] if tr is None else [tr]), dim=1).repeat(bs, 1)

    x_data = torch.cat([x.permute(0, 2, 3, 1, 4).reshape(bs * g_, t * v), node_attrs(data)], dim=1).to(dtype)
    x_hidden = node_attrs(hidden).to(dtype)
    n_data, n_hidden = x_data.shape[0], x_hidden.shape[0]
    plans, attrs = {}, {}
    for key, mod, (ns, nd) in (("enc", "encoder", (n_data, n_hidden)), ("proc", "processor", (n_hidden, n_hidden)),
                               ("dec", "decoder", (n_hidden, n_data))):
        ei = graph[f"{key}_edge_index"]
        if ei.device != x.device:
            ei = ei.to(x.device)
        inc = torch.tensor([[ns // bs], [nd // bs]], dtype=ei.dtype, device=ei.device) if bs > 1 else None
        if plan_cache is not None:  # keyed on the edge-index tensor's identity / version and the batch size
            plans[key] = plan_cache.get(ei, ns, nd, bs, inc)
        else:  # batched graph: sample i's edges are shifted by i * (nodes per sample) (layers/mapper.py:150-171)
            plans[key] = runtime.build_edge_plan(runtime.expand_edges(ei, inc, bs) if bs > 1 else ei, ns, nd)
        # every edge set has its own attribute width (its trainable tensor may be absent or of another size): the
        # folded width ``up`` = attributes + the constant-1 column, rounded to the kernel's 4-float granule
        trainable = sd.get(f"{mod}.trainable.trainable")
        width = graph[f"{key}_edge_attr"].shape[1] + (0 if trainable is None else trainable.shape[1])
        attrs[key] = _edge_attr_csr(graph[f"{key}_edge_attr"].to(x.device), trainable, plans[key],
                                    ops.round_up(width + 1, 4), bs)

    xs = linear(x_data, sd["encoder.emb_nodes_src.weight"], sd["encoder.emb_nodes_src.bias"])
    xd = linear(x_hidden, sd["encoder.emb_nodes_dst.weight"], sd["encoder.emb_nodes_dst.bias"])
    x_latent = gt_mapper_block(xs, xd, sd, "encoder.proc", attrs["enc"], plans["enc"], num_heads, act)
    x_proc = x_latent
    per_chunk = num_layers // num_chunks
    for ci in range(num_chunks):
        for bi in range(per_chunk):
            x_proc = gt_processor_block(x_proc, sd, f"processor.proc.{ci}.blocks.{bi}", attrs["proc"], plans["proc"],
                                        num_heads, act)
    x_latent_proc = x_proc + x_latent
    xg = linear(x_data, sd["decoder.emb_nodes_dst.weight"], sd["decoder.emb_nodes_dst.bias"])
    out = gt_mapper_block(x_latent_proc, xg, sd, "decoder.proc", attrs["dec"], plans["dec"], num_heads, act)
    out = layer_norm(out, sd["decoder.node_data_extractor.0.weight"], sd["decoder.node_data_extractor.0.bias"])
    out = linear(out, sd["decoder.node_data_extractor.1.weight"], sd["decoder.node_data_extractor.1.bias"])
    y = out.float().reshape(b, ens, g_, -1).clone()
    y[..., list(prognostic_out)] = y[..., list(prognostic_out)] + x[:, -1, :, :, list(prognostic_in)]
    return y
