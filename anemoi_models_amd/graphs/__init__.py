"""Graph containers and the deterministic synthetic graph generator (host side)."""

from .data import GraphData  # noqa: F401
from .data import Store  # noqa: F401
