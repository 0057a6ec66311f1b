"""``DotDict`` and ``instantiate``: what the model root needs from anemoi.utils / hydra when they are absent.

If hydra is installed its ``instantiate`` is used (so real anemoi-training configs behave exactly as
upstream); otherwise a minimal re-implementation resolves ``_target_`` and merges keyword arguments.
``_target_`` strings naming the reference package (``anemoi.models.layers...`` / ``anemoi.models.models...``)
are redirected to this package, so an unmodified anemoi-training config instantiates the MI355X classes.
"""

from __future__ import annotations

import importlib
from typing import Any

REFERENCE_PREFIX = "anemoi.models."
PACKAGE_PREFIX = "anemoi_models_amd."


class DotDict(dict):
    """Dictionary with attribute access, recursively applied to nested dicts and lists of dicts."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        for k, v in list(self.items()):
            self[k] = self._wrap(v)

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, DotDict):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(i) for i in v)
        return v

    def __getattr__(self, key: str) -> Any:
        try:
            return self[key]
        except KeyError as e:
            raise AttributeError(key) from e

    def __setattr__(self, key: str, value: Any) -> None:
        self[key] = self._wrap(value)


def resolve_target(target: str):
    if target.startswith(REFERENCE_PREFIX):
        target = PACKAGE_PREFIX + target[len(REFERENCE_PREFIX):]
    module, name = target.rsplit(".", 1)
    return getattr(importlib.import_module(module), name)


def instantiate(config, *args, **kwargs):
    """Create ``config["_target_"](*args, **config_without_meta_keys, **kwargs)``."""
    cfg = dict(config)
    target = cfg.pop("_target_")
    for meta in ("_convert_", "_recursive_", "_partial_"):
        cfg.pop(meta, None)
        kwargs.pop(meta, None)
    cfg.update(kwargs)
    return resolve_target(target)(*args, **cfg)
