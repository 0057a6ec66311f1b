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


def test_block_level_entry_points_validate_their_argument_block_without_gpu():
    """anemoi_gt_block_tail / anemoi_gt_processor_block_forward (SURVEY section 8b's block-level boundary): the ctypes
    mirror of ``anemoi_gt_block_args`` has the size the library was compiled with, and a bad block is refused with a
    status code before anything is launched."""
    from anemoi_models_amd import _lib

    lib = _lib.load()
    a = _lib.GtBlockArgs()
    assert lib.anemoi_gt_block_tail(None, None) == _lib.ANEMOI_ERR_INVALID and b"null argument" in lib.anemoi_last_error()
    a.struct_bytes = ctypes.sizeof(_lib.GtBlockArgs) - 8
    assert lib.anemoi_gt_processor_block_forward(ctypes.byref(a), None) == _lib.ANEMOI_ERR_INVALID
    assert b"out of sync" in lib.anemoi_last_error()
    a.struct_bytes = ctypes.sizeof(_lib.GtBlockArgs)  # the layout check passes: both sides agree on sizeof
    a.dtype = _lib.F32
    assert lib.anemoi_gt_processor_block_forward(ctypes.byref(a), None) == _lib.ANEMOI_ERR_UNSUPPORTED
    assert b"bf16" in lib.anemoi_last_error()
    a.dtype, a.n_dst, a.C, a.H, a.up, a.hidden, a.k_proj, a.ld_att = _lib.BF16, 16, 64, 4, 4, 128, 64, 128
    assert lib.anemoi_gt_block_tail(ctypes.byref(a), None) == _lib.ANEMOI_ERR_INVALID  # K of the projection < C + H * up
    assert b"projection K" in lib.anemoi_last_error()


def test_block_level_argument_blocks_are_checked_without_gpu():
    """The block-level entry points refuse an argument block of another layout version / with missing pointers before
    anything is launched (struct_bytes handshake across the FFI)."""
    import ctypes

    from anemoi_models_amd import _lib

    lib = _lib.load()
    a = _lib.TfmBlockArgs()
    a.struct_bytes = 8
    assert lib.anemoi_transformer_block_forward(ctypes.byref(a), None) == _lib.ANEMOI_ERR_INVALID
    assert b"out of sync" in lib.anemoi_last_error()
    a.struct_bytes = ctypes.sizeof(_lib.TfmBlockArgs)
    a.dtype, a.rows, a.B, a.S, a.C, a.H, a.hidden = 1, 64, 1, 64, 128, 4, 512
    assert lib.anemoi_transformer_block_forward(ctypes.byref(a), None) == _lib.ANEMOI_ERR_INVALID
    assert b"null pointer" in lib.anemoi_last_error()
    g = _lib.GtBlockArgs()
    g.struct_bytes = ctypes.sizeof(_lib.GtBlockArgs) - 8
    assert lib.anemoi_gt_block_tail(ctypes.byref(g), None) == _lib.ANEMOI_ERR_INVALID
