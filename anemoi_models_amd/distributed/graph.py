"""Import location of the reference's graph collectives (distributed/graph.py:19-137); implemented in ``collectives.py``."""

from .collectives import gather_tensor  # noqa: F401
from .collectives import reduce_shard_tensor  # noqa: F401
from .collectives import reduce_tensor  # noqa: F401
from .collectives import shard_tensor  # noqa: F401
from .collectives import sync_tensor  # noqa: F401
