#!/usr/bin/env python
"""The reference's OWN layer / model tests against this package (build container only; needs /root/reference).

    python tools/run_reference_tests.py [pytest arguments]

* the third-party packages the reference's tests import and this image lacks (torch_geometric, hydra, anemoi.utils) come
  from the stand-ins of tests/golden/_ref_stubs.py -- ``HeteroData`` and ``DotDict`` are all the tests need of them;
* ``anemoi.models`` (and every sub-module) is ALIASED to ``anemoi_models_amd``: the reference's tests construct this
  package's classes through the reference's import paths, constructor arguments and forward signatures;
* there is no GPU here and ``anemoi_models_amd.ops`` has no CPU implementation, so the ops are the CPU stand-ins of
  tests/_cpu_ops.py (plain torch + the oracle) -- what this run checks is the API surface (names, kwargs, shapes, error
  types, ``state_dict`` behaviour, the chunk / shard protocol), not the kernels;
* the test files are run where they lie (``/root/reference/tests/{layers,models}``); nothing of them is copied.

Not part of the GPU-box runs (``/root/reference`` does not exist there).  Prints pytest's summary; exit code = pytest's."""
import importlib
import os
import pkgutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def main() -> int:
    if not os.path.isdir(os.path.join(REF, "tests")):
        print(f"{REF}/tests not found: this tool runs in the build container only")
        return 2
    import pytest
    import _ref_stubs

    _ref_stubs.install(os.path.join(ROOT, "tests", "golden", "_no_reference_src"))  # stand-ins only, NOT the reference sources
    sys.path[:] = [p for p in sys.path if "_no_reference_src" not in p]
    import anemoi_models_amd as pkg

    # alias: anemoi, anemoi.models and every sub-module of this package under the reference's names
    import types

    top = sys.modules.get("anemoi") or types.ModuleType("anemoi")
    top.__path__ = []
    sys.modules["anemoi"] = top
    sys.modules["anemoi.models"] = pkg
    top.models = pkg
    top.utils = sys.modules["anemoi.utils"]
    for info in pkgutil.walk_packages(pkg.__path__, pkg.__name__ + "."):
        if any(part.startswith("_") for part in info.name.split(".")[1:]) or ".csrc" in info.name or ".lib" in info.name:
            continue
        try:
            mod = importlib.import_module(info.name)
        except Exception as exc:  # noqa: BLE001
            print(f"(not aliased: {info.name}: {type(exc).__name__}: {exc})")
            continue
        sys.modules["anemoi.models" + info.name[len(pkg.__name__):]] = mod

    class _Plugin:  # the CPU stand-ins of the ops for every test (monkeypatch scope = the test)
        @pytest.fixture(autouse=True)
        def _cpu_ops(self, monkeypatch):
            import _cpu_ops

            _cpu_ops.install(monkeypatch)
            yield

    args = sys.argv[1:] or ["-q", "-p", "no:cacheprovider", os.path.join(REF, "tests", "layers"), os.path.join(REF, "tests", "models")]
    return pytest.main(args, plugins=[_Plugin()])


if __name__ == "__main__":
    sys.exit(main())
