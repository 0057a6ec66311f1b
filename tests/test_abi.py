"""The C-ABI library loads without a GPU and exports every symbol that include/anemoi_amd.h declares."""

import ctypes
import os
import re

from conftest import ROOT


def declared_functions():
    text = open(os.path.join(ROOT, "include", "anemoi_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(anemoi_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_bound_and_exported():
    from anemoi_models_amd import _build, _lib

    _build.build()  # cross-compiles for gfx950 if the library is missing or stale; no GPU needed
    names = declared_functions()
    assert "anemoi_gt_edge_attention_folded" in names and "anemoi_linear" in names and len(names) >= 11
    assert sorted(_lib.SIGNATURES) == names, "ctypes signature table out of sync with include/anemoi_amd.h"
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} missing from libanemoi_amd.so"
    lib = _lib.load()
    assert lib.anemoi_abi_version() == _lib.ABI_VERSION
    assert lib.anemoi_last_error() is not None


def test_argument_validation_without_gpu():
    """Status codes and messages come back through the ABI before anything is launched."""
    from anemoi_models_amd import _lib

    lib = _lib.load()
    st = lib.anemoi_layer_norm(0, None, 0, None, None, None, 0, 4, 8, 1e-5, None)
    assert st == _lib.ANEMOI_ERR_INVALID and b"null pointer" in lib.anemoi_last_error()
    st = lib.anemoi_linear(1, 1, 16, 48, 16, None, None, 0, 16, 8, 4, 8, 48, 0, None)  # K = 48: not slab padded
    assert st == _lib.ANEMOI_ERR_INVALID and b"multiple" in lib.anemoi_last_error()
