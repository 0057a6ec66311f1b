"""Model-parallel execution of the forward path: mesh node-partition + halo all-to-all-v (RCCL over xGMI).

The reference shards one model instance with Ulysses-style head <-> sequence all-to-alls of full ``[N, C]`` / ``[E, C]``
tensors, 5 per block (reference distributed/transformer.py:21-82, layers/block.py:389-392,411).  Here each rank owns a
contiguous range of the (Morton-ordered) mesh, computes everything row-local for its range and exchanges only the k|v
rows of boundary ("halo") nodes once per block.  The external contract is unchanged: every rank of the model group
passes the full input and receives the full output.
"""

from .shapes import change_channels_in_shape  # noqa: F401
from .shapes import get_shape_shards  # noqa: F401
