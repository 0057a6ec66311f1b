"""Small wrappers mirroring reference layers/utils.py:16-39."""

from __future__ import annotations

from torch import Tensor
from torch import nn
from torch.utils.checkpoint import checkpoint


class CheckpointWrapper(nn.Module):
    """Runs ``module`` under non-reentrant activation checkpointing (reference layers/utils.py:16-24)."""

    def __init__(self, module: nn.Module) -> None:
        super().__init__()
        self.module = module

    def forward(self, *args, **kwargs):
        from .. import runtime

        dd = runtime.device_dropout()
        if dd is None:
            return checkpoint(self.module, *args, **kwargs, use_reentrant=False)
        frozen = dd.pinned()  # the recomputation re-enters the forward's dropout step (training._checkpoint)

        def pinned(*a, **k):
            with frozen:
                return self.module(*a, **k)

        return checkpoint(pinned, *args, **kwargs, use_reentrant=False)


class AutocastLayerNorm(nn.LayerNorm):
    """LayerNorm whose result is cast back to the input dtype (reference layers/utils.py:27-39)."""

    def forward(self, x: Tensor) -> Tensor:
        return super().forward(x).type_as(x)
