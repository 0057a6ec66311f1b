"""Import location of the reference's ``get_memory_format`` (distributed/utils.py); implemented in ``collectives.py``."""

from .collectives import get_memory_format  # noqa: F401
