"""Differentiable Linear / LayerNorm on the HIP kernels -- the dense half of the backward pass (first
step; the edge-attention backward is the other half and not in this build).

``linear(x, weight, bias, act, residual)`` and ``layer_norm(x, gamma, beta, eps)`` are ``torch.autograd.Function``s
whose forward is the inference path's fused GEMM / LayerNorm kernel and whose backward runs on the same GEMM kernels:

* ``dX = dpre @ W``            -> ``ops.linear(dpre, W^T)``            (``W^T`` by ``anemoi_transpose``)
* ``dW = dpre^T @ X`` (f32)    -> ``ops.linear(dpre^T, X^T, f32 out)`` (reduction over the rows, zero padded to K slabs)
* ``db = sum_rows dpre``       -> ``anemoi_col_sum``
* ``dpre = dy * act'(pre)``    -> ``anemoi_act_backward`` (the pre-activation is saved: one extra GEMM output)
* LayerNorm                    -> ``anemoi_layer_norm_backward`` from the forward's row statistics

What torch derives for the reference's ``nn.Linear`` / ``nn.GELU`` / ``nn.LayerNorm`` (layers/block.py:504-508, 631-633,
layers/mlp.py:74-84) when anemoi-training calls ``.backward()``.  Parameters stay f32 (their gradients too); activations
and activation gradients are in the compute dtype.  The model classes do not use these yet (``runtime.require_inference``):
training needs the edge-phase backward as well.
"""

from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

from . import ops


def _pack(w: Tensor, dtype: torch.dtype) -> Tensor:
    """``[N, K]`` f32 parameter -> compute dtype with K zero-padded to the GEMM's slab multiple."""
    n, k = w.shape
    kp = ops.round_up(k, ops.k_multiple(dtype))
    if kp == k and w.dtype == dtype and w.is_contiguous():
        return w
    out = torch.zeros((n, kp), dtype=dtype, device=w.device)
    out[:, :k] = w.detach().to(dtype)
    return out


_TORCH_ACT = {"GELU": torch.nn.functional.gelu, "SiLU": torch.nn.functional.silu, "ReLU": torch.relu}


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias: Optional[Tensor], act: str, residual: Optional[Tensor]):
        dtype = x.dtype
        k = weight.shape[1]
        kp = ops.round_up(k, ops.k_multiple(dtype))
        if x.shape[1] not in (k, kp):
            raise ValueError(f"linear: x has {x.shape[1]} columns, weight expects {k}")
        xk = x if x.shape[1] == kp else ops.convert_pad(x, dtype, kp)
        w = _pack(weight, dtype)
        b = None if bias is None else bias.detach().float().contiguous()
        if act == "Identity":
            pre = None
            y = ops.linear(xk, w, b, residual=residual)
        else:
            # the pre-activation is an output of its own here (the backward needs act'(pre)); the activation and the
            # residual then run on the stored value instead of in the GEMM epilogue
            pre = ops.linear(xk, w, b)
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
        dx = dw = db = dres = None
        if ctx.needs_input_grad[0]:
            # dX [M, K] = dpre [M, N] @ W [N, K]: a Linear whose weight is W^T [K, N] (N is the reduction dimension)
            np_ = ops.round_up(n, kmul)
            wt = ops.transpose(weight.detach().to(dtype).contiguous(), ld_out=np_)
            dp = dpre if n == np_ else ops.convert_pad(dpre, dtype, np_)
            dx = ops.linear(dp, wt)
            if ctx.x_cols != k:
                dx = ops.convert_pad(dx, dtype, ctx.x_cols)
        if ctx.needs_input_grad[1]:
            # dW [N, K] = dpre^T [N, M] @ X [M, K]: a Linear with x' = dpre^T, weight' = X^T, reduction over the M rows
            mp = ops.round_up(dpre.shape[0], kmul)
            xt = ops.transpose(xk[:, :k] if xk.shape[1] != k else xk, ld_out=mp)
            dw = ops.linear(ops.transpose(dpre, ld_out=mp), xt, out_dtype=torch.float32).to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = ops.col_sum(dpre)
        if ctx.has_res and ctx.needs_input_grad[4]:
            dres = dy
        return dx, dw, db, None, dres


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, gamma: Tensor, beta: Tensor, eps: float):
        stats = ops.row_stats(x, eps)
        ctx.save_for_backward(x, stats, gamma)
        return ops.layer_norm(x, gamma.detach().float().contiguous(), beta.detach().float().contiguous(), eps)

    @staticmethod
    def backward(ctx, dy: Tensor):
        x, stats, gamma = ctx.saved_tensors
        dx, dgamma, dbeta = ops.layer_norm_backward(x, stats, gamma, dy.contiguous())
        return dx, dgamma.to(gamma.dtype), dbeta.to(gamma.dtype), None


def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None, act: str = "Identity",
           residual: Optional[Tensor] = None) -> Tensor:
    """``act(x @ weight.T + bias) + residual`` with gradients for ``x``, ``weight``, ``bias`` and ``residual``.
    ``x`` / ``residual`` in the compute dtype (f32 or bf16), ``weight [N, K]`` / ``bias [N]`` f32 parameters."""
    return _Linear.apply(x, weight, bias, act, residual)


def layer_norm(x: Tensor, gamma: Tensor, beta: Tensor, eps: float = 1e-5) -> Tensor:
    """``LayerNorm(x) * gamma + beta`` over the last dimension with gradients for ``x``, ``gamma``, ``beta``."""
    return _LayerNorm.apply(x, gamma, beta, eps)
