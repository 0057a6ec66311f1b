"""MLP mirror (reference layers/mlp.py:22-89) executed with the fused Linear(+activation) and LayerNorm kernels.

``state_dict`` keys are those of the reference: ``model.<i>.{weight,bias}`` for the Linear layers at the
even positions of the ``nn.Sequential`` and the trailing ``AutocastLayerNorm``.
"""

from __future__ import annotations

import logging
from typing import Optional

import torch
from torch import Tensor
from torch import nn

from .. import ops
from .. import runtime
from .utils import AutocastLayerNorm
from .utils import CheckpointWrapper

LOGGER = logging.getLogger(__name__)


FUSED_ACTIVATIONS = ("Identity", "GELU", "SiLU", "ReLU")  # epilogues of the fused Linear kernel


def fused_activation_name(module: nn.Module) -> Optional[str]:
    """Name of the GEMM epilogue that computes ``module``, or None when it has to run as a torch op behind the Linear
    (the reference takes any ``torch.nn`` activation by name, layers/mlp.py:66-72; e.g. ``Tanh`` in its block tests)."""
    name = type(module).__name__
    if name not in FUSED_ACTIVATIONS:
        return None
    if name == "GELU" and getattr(module, "approximate", "none") != "none":
        return None  # the kernel's GELU is the exact erf form
    return name


def activation_class(name: str):
    """``getattr(nn, name)`` with the reference's error behaviour (layers/mlp.py:66-72, layers/block.py:75-79)."""
    try:
        return getattr(nn, name)
    except AttributeError as ae:
        LOGGER.error("Activation function %s not supported", name)
        raise RuntimeError from ae


class NativeSequential:
    """Runs an ``nn.Sequential`` of Linear / activation / LayerNorm modules on the HIP kernels.

    Each ``Linear`` is fused with the activation that follows it; weights are packed (cast + K padded)
    once per dtype and cached.  ``residual`` is added by the epilogue of the last Linear when the
    sequence does not end in a LayerNorm, otherwise by a separate add.
    """

    def __init__(self, seq: nn.Sequential) -> None:
        self.seq = seq
        self.cache = runtime.PackedWeights()
        self.steps = []
        mods = list(seq)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.Linear):
                act = "Identity"
                if i + 1 < len(mods) and not isinstance(mods[i + 1], (nn.Linear, nn.LayerNorm)):
                    fused = fused_activation_name(mods[i + 1])
                    if fused is not None:
                        act = fused
                        i += 1
                self.steps.append(("linear", m, act))
            elif isinstance(m, nn.LayerNorm):
                self.steps.append(("ln", m, None))
            else:  # an activation without a GEMM epilogue (Tanh, LeakyReLU, ...): a torch op on the Linear's result
                self.steps.append(("act", m, None))
            i += 1

    def folded_pair(self, dtype: torch.dtype):
        """The packed operands of ``LayerNorm -> Linear + act -> Linear`` (the node MLP of a GraphTransformer block) for
        the block-level C entry points: ``(eps, act, w1', b1', colsum1, w2, b2)`` -- the SAME cached tensors ``__call__``
        hands to ``ops.linear`` --, or ``None`` when this sequence has another shape or the LayerNorm fold is off."""
        kinds = [s[0] for s in self.steps]
        if kinds != ["ln", "linear", "linear"] or not runtime.ln_fold_enabled(dtype):
            return None
        (_, ln, _), (_, m1, act), (_, m2, act2) = self.steps
        if act2 != "Identity" or m1.in_features % ops.k_multiple(dtype) != 0 or m2.in_features % ops.k_multiple(dtype) != 0:
            return None
        wf, bf, cs = self.cache.get(("lnfold", 1, dtype), [m1.weight, m1.bias, ln.weight, ln.bias],
                                    lambda: runtime.fold_layer_norm(m1.weight.detach().float(), m1.bias, ln.weight, ln.bias,
                                                                    dtype))
        w2 = self.cache.get(("w", 2, dtype), [m2.weight], lambda: runtime.pack_weight([m2.weight], dtype))
        b2 = None if m2.bias is None else runtime.f32c(m2.bias)
        return ln.eps, act, wf, bf, cs, w2, b2

    def __call__(self, x: Tensor, residual: Optional[Tensor] = None, out_dtype: Optional[torch.dtype] = None,
                 start: int = 0, out_stats_eps: Optional[float] = None) -> Tensor:
        """Run steps ``start..`` (``start`` > 0: the caller has already produced the output of the earlier steps).
        ``out_stats_eps``: the result feeds a LayerNorm with this epsilon next -- the last Linear's epilogue produces
        its row statistics (``ops.linear(stats_eps=...)``)."""
        dtype = x.dtype
        last_linear = max(i for i, s in enumerate(self.steps) if s[0] == "linear")
        ends_with_ln = self.steps[-1][0] != "linear"  # something (LayerNorm / torch activation) follows the last Linear
        pending_ln = None  # a LayerNorm whose consumer is the next Linear: folded into it (row_stats + linear_ln)
        for i, (kind, m, act) in enumerate(self.steps):
            if i < start:
                continue
            if kind == "linear":
                fuse_res = residual is not None and i == last_linear and not ends_with_ln
                last = i == len(self.steps) - 1
                kw = dict(act=act, residual=residual if fuse_res else None, out_dtype=out_dtype if last else None)
                if last and out_stats_eps is not None and (residual is None or fuse_res):
                    kw["stats_eps"] = out_stats_eps
                if pending_ln is not None:
                    ln, stats = pending_ln
                    pending_ln = None
                    wf, bf, cs = self.cache.get(("lnfold", i, dtype), [m.weight, m.bias, ln.weight, ln.bias],
                                                lambda m=m, ln=ln: runtime.fold_layer_norm(
                                                    m.weight.detach().float(), m.bias, ln.weight, ln.bias, dtype))
                    x = ops.linear(x, wf, bf, ln=(stats, cs), **kw)
                    continue
                w = self.cache.get(("w", i, dtype), [m.weight], lambda m=m: runtime.pack_weight([m.weight], dtype))
                b = None if m.bias is None else runtime.f32c(m.bias)
                if x.shape[1] != w.shape[1]:
                    x = ops.convert_pad(x, dtype, w.shape[1])
                x = ops.linear(x, w, b, **kw)
            elif kind == "act":
                x = m(x)
            else:
                nxt = self.steps[i + 1] if i + 1 < len(self.steps) else None
                if (nxt is not None and nxt[0] == "linear" and runtime.ln_fold_enabled(dtype)
                        and x.shape[1] % ops.k_multiple(dtype) == 0 and nxt[1].in_features == x.shape[1]):
                    pending_ln = (m, ops.row_stats(x, m.eps))
                else:
                    fuse = residual is not None and ends_with_ln and i == len(self.steps) - 1 and residual.dtype == x.dtype
                    x = ops.layer_norm(x, runtime.f32c(m.weight), runtime.f32c(m.bias), m.eps,
                                       residual=residual if fuse else None)  # trailing LayerNorm + skip connection: one pass
                    if fuse:
                        residual = None
        if residual is not None and ends_with_ln:
            x = ops.add(x, residual)
        return x


class MLP(nn.Module):
    """Linear, act, (Linear, act) x (n_extra_layers + 1), Linear, [act], [LayerNorm]."""

    def __init__(
        self,
        in_features: int,
        hidden_dim: int,
        out_features: int,
        n_extra_layers: int = 0,
        activation: str = "SiLU",
        final_activation: bool = False,
        layer_norm: bool = True,
        checkpoints: bool = False,
    ) -> None:
        super().__init__()
        act = activation_class(activation)
        layers = [nn.Linear(in_features, hidden_dim), act()]
        for _ in range(n_extra_layers + 1):
            layers += [nn.Linear(hidden_dim, hidden_dim), act()]
        layers.append(nn.Linear(hidden_dim, out_features))
        if final_activation:
            layers.append(act())
        if layer_norm:
            layers.append(AutocastLayerNorm(out_features))
        seq = nn.Sequential(*layers)
        self.model = CheckpointWrapper(seq) if checkpoints else seq
        self._native: Optional[NativeSequential] = None

    def native(self) -> NativeSequential:
        if self._native is None:
            seq = self.model.module if isinstance(self.model, CheckpointWrapper) else self.model
            self._native = NativeSequential(seq)
        return self._native

    def forward(self, x: Tensor) -> Tensor:
        from .. import training

        if x.dim() != 2:  # nn.Linear semantics: any leading dimensions (reference tests/layers/test_mlp.py: [B, N, F])
            lead = x.shape[:-1]
            y = self.forward(x.reshape(-1, x.shape[-1]))
            return y.reshape(*lead, y.shape[-1])
        if training.wants_grad(self, x):
            return training.mlp(self, training._cast(x, runtime.compute_dtype(x)))
        dtype = runtime.compute_dtype(x)
        xin = x if x.dtype == dtype else x.to(dtype)
        return self.native()(xin.contiguous() if xin.stride(-1) != 1 else xin)


def linear_native(cache: runtime.PackedWeights, tag: str, lin: nn.Linear, x: Tensor, act: str = "Identity",
                  residual: Optional[Tensor] = None, out_dtype: Optional[torch.dtype] = None,
                  stats_eps: Optional[float] = None) -> Tensor:
    """One ``nn.Linear`` on the fused GEMM kernel; ``x`` may carry zero K-padding or none (it is padded here)."""
    dtype = x.dtype
    w = cache.get((tag, "w", dtype), [lin.weight], lambda: runtime.pack_weight([lin.weight], dtype))
    b = None if lin.bias is None else runtime.f32c(lin.bias)
    if x.shape[1] > w.shape[1] and x.shape[1] % ops.k_multiple(dtype) == 0:
        # wider K padding than the weight needs (the columns behind in_features meet zero weights whatever they hold)
        kp = x.shape[1]
        w = cache.get((tag, "w", dtype, kp), [lin.weight], lambda: runtime.pack_weight([lin.weight], dtype, k_pad=kp))
    if x.shape[1] != w.shape[1]:
        if x.shape[1] != lin.in_features:
            raise ValueError(f"{tag}: input has {x.shape[1]} features, expected {lin.in_features}")
        x = ops.convert_pad(x, dtype, w.shape[1])
    return ops.linear(x, w, b, act=act, residual=residual, out_dtype=out_dtype, stats_eps=stats_eps)
