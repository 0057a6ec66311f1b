"""CPU oracle for the anemoi-models encoder-processor-decoder forward path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``anemoi_models_amd/`` (the product)
may import this package.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker / the
reported CPU baseline -- never as the thing that is shipped or measured as the
MI355X path.

The oracle is a plain-PyTorch (CPU, fp32) restatement of the reference's
algorithm for the hot path.  Each function cites the
reference ``file:line`` it follows.  The sparse arithmetic the reference
delegates to ``torch-geometric>=2.3,<2.5`` (pyproject.toml:49; un-vendored, not
installed here) is restated in :mod:`oracle.pyg_semantics` from PyG's published
contract.

Pinning: ``tests/golden/make_golden.py`` imports the real reference sources
from ``/root/reference/src`` (through small stand-ins for the missing
third-party packages) in the build container, runs them on seeded inputs and
stores inputs + outputs under ``tests/golden/``.  ``tests/test_oracle_golden.py``
checks this oracle against those vectors.  The reference's own tests hold no
golden vectors for this path, so the fixtures generated
from the reference itself are the pin.
"""

from . import pyg_semantics  # noqa: F401
from . import reference_path  # noqa: F401
