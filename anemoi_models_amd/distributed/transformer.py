"""Import location of the reference's head / sequence all-to-alls (distributed/transformer.py:85-130); implemented in
``collectives.py``."""

from .collectives import shard_heads  # noqa: F401
from .collectives import shard_sequence  # noqa: F401
