"""ctypes binding of ``libanemoi_amd.so`` -- the C ABI declared in ``include/anemoi_amd.h``.

There is no CPU fallback: if the library is missing or a kernel call fails the caller gets an
exception.  The product path never routes through ``oracle/``.
"""

from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p
from ctypes import c_float
from ctypes import c_int
from ctypes import c_int64
from ctypes import c_uint32
from ctypes import c_void_p

import torch  # noqa: F401  -- FIRST: puts torch's bundled HIP runtime (libamdhip64.so.7) in the process so that
#                        this library binds to the same runtime instance that owns torch's device memory

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, "lib", "libanemoi_amd.so")

ANEMOI_OK = 0
ANEMOI_ERR_INVALID = 1
ANEMOI_ERR_UNSUPPORTED = 2
ANEMOI_ERR_LAUNCH = 3

F32 = 0
BF16 = 1

ACT_NONE, ACT_GELU, ACT_SILU, ACT_RELU = 0, 1, 2, 3
ACT_CODES = {"Identity": ACT_NONE, "GELU": ACT_GELU, "SiLU": ACT_SILU, "ReLU": ACT_RELU}

ABI_VERSION = 41


class GtBlockArgs(ctypes.Structure):
    """``anemoi_gt_block_args`` of include/anemoi_amd.h (block-level entry points), field for field."""

    _fields_ = [
        ("struct_bytes", c_int64), ("n_dst", c_int64),
        ("dtype", ctypes.c_int32), ("C", ctypes.c_int32), ("H", ctypes.c_int32), ("up", ctypes.c_int32),
        ("hidden", ctypes.c_int32), ("act", ctypes.c_int32), ("k_proj", ctypes.c_int32), ("n_in", ctypes.c_int32),
        ("eps_mlp", c_float), ("eps_out", c_float),
        ("x", c_void_p), ("ldx", c_int64), ("x_stats", c_void_p),
        ("w_in", c_void_p), ("b_in", c_void_p), ("cs_in", c_void_p),
        ("sq", c_void_p), ("ld_sq", c_int64),
        ("q", c_void_p), ("k", c_void_p), ("v", c_void_p), ("x_r", c_void_p), ("u", c_void_p),
        ("ldq", c_int64), ("ldkv", c_int64), ("ldr", c_int64), ("ldu", c_int64),
        ("edge_attr", c_void_p), ("rowptr", c_void_p), ("col", c_void_p),
        ("att", c_void_p), ("ld_att", c_int64),
        ("w_proj", c_void_p), ("b_proj", c_void_p), ("res", c_void_p), ("ld_res", c_int64), ("y", c_void_p),
        ("y_stats", c_void_p),
        ("w_fc1", c_void_p), ("b_fc1", c_void_p), ("cs_fc1", c_void_p), ("h", c_void_p),
        ("w_fc2", c_void_p), ("b_fc2", c_void_p), ("out", c_void_p), ("out_stats", c_void_p),
        ("stats_ws", c_void_p), ("stats_ws_bytes", c_int64),
        ("run_ptr", c_void_p), ("run_perm", c_void_p), ("n_runs", c_int64),
        ("sched", c_void_p), ("sched_slots", ctypes.c_int32), ("sched_steps", ctypes.c_int32), ("n_src", c_int64),
        ("run_dst", c_void_p), ("n_edges", c_int64),
        ("tile_hdr", c_void_p), ("tile_dst", c_void_p), ("tile_src", c_void_p), ("tile_slot", c_void_p), ("tile_xcd", c_void_p),
        ("tile_max_per_xcd", ctypes.c_int32), ("tile_src_cap", ctypes.c_int32), ("tile_edge_cap", ctypes.c_int32),
        ("tile_pad", ctypes.c_int32),
    ]

class TfmBlockArgs(ctypes.Structure):
    """``anemoi_tfm_block_args`` of include/anemoi_amd.h (``anemoi_transformer_block_forward``), field for field."""

    _fields_ = [
        ("struct_bytes", c_int64), ("rows", c_int64),
        ("dtype", ctypes.c_int32), ("B", ctypes.c_int32), ("S", ctypes.c_int32), ("C", ctypes.c_int32), ("H", ctypes.c_int32),
        ("hidden", ctypes.c_int32), ("act", ctypes.c_int32), ("window", ctypes.c_int32),
        ("eps1", c_float), ("eps2", c_float),
        ("dropout_p", c_float), ("dropout_seed", c_uint32), ("dropout_seed_dev", c_void_p),
        ("dropout_h0", ctypes.c_int32), ("dropout_h_total", ctypes.c_int32),
        ("x", c_void_p), ("ldx", c_int64),
        ("ln1_w", c_void_p), ("ln1_b", c_void_p), ("ln2_w", c_void_p), ("ln2_b", c_void_p),
        ("w_qkv", c_void_p), ("b_qkv", c_void_p), ("w_proj", c_void_p), ("b_proj", c_void_p),
        ("w_fc1", c_void_p), ("b_fc1", c_void_p), ("w_fc2", c_void_p), ("b_fc2", c_void_p),
        ("h_ln", c_void_p), ("qkv", c_void_p), ("att", c_void_p), ("y", c_void_p), ("h", c_void_p),
        ("mhsa_ws", c_void_p), ("out", c_void_p),
    ]


# name -> (restype, argtypes); must list every symbol of include/anemoi_amd.h (checked by tests/test_abi.py)
SIGNATURES = {
    "anemoi_abi_version": (c_int, []),
    "anemoi_build_info": (ctypes.c_char_p, []),
    "anemoi_last_error": (c_char_p, []),
    "anemoi_layer_norm": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int,
                                  c_float, c_void_p]),
    "anemoi_layer_norm_residual": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                           c_int64, c_int, c_float, c_void_p]),
    "anemoi_layer_norm_stats": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                        c_int, c_float, c_void_p]),
    "anemoi_linear": (c_int, [c_int, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                              c_int64, c_int64, c_int, c_int, c_int, c_void_p]),
    "anemoi_linear_stats": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                    c_void_p, c_int64, c_int64, c_int, c_int, c_void_p, c_int64, c_float, c_void_p,
                                    c_void_p]),
    "anemoi_row_stats": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_int64, c_int, c_float, c_void_p]),
    "anemoi_linear_ln": (c_int, [c_int, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_void_p]),
    "anemoi_edge_attr_csr": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_void_p, c_void_p, c_int, c_int,
                                     c_int64, c_void_p]),
    "anemoi_gt_edge_attention_folded": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p,
                                                c_int64, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_void_p,
                                                c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "anemoi_gt_edge_attention_folded_runs": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p,
                                                     c_int64, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_void_p,
                                                     c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int,
                                                     c_int, c_void_p]),
    "anemoi_gt_edge_attention_folded_groups": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p,
                                                       c_int64, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_void_p,
                                                       c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64,
                                                       c_void_p, c_int64, c_int, c_int, c_void_p]),
    "anemoi_edge_schedule_shape": (c_int, [c_int, c_int64, c_int, c_void_p, c_void_p]),
    "anemoi_gt_edge_attention_folded_sched": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p,
                                                      c_int64, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_void_p,
                                                      c_void_p, c_int, c_int, c_int64, c_int64, c_void_p, c_int64, c_void_p,
                                                      c_int64, c_int, c_int, c_void_p]),
    "anemoi_gt_edge_attention_folded_tiles": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p,
                                                      c_int64, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_void_p,
                                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                                      c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int,
                                                      c_void_p]),
    "anemoi_linear_dual": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                   c_int64, c_int, c_int, c_int, c_void_p]),
    "anemoi_linear_actgrad": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64,
                                      c_int, c_int, c_int, c_void_p]),
    "anemoi_gt_conv": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                               c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_float, c_uint32,
                               c_void_p, c_void_p]),
    "anemoi_gt_conv_backward_dst": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                            c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_int64, c_int64, c_int, c_int, c_float, c_uint32, c_void_p,
                                            c_void_p]),
    "anemoi_gt_conv_backward_src": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                            c_int64, c_int, c_int, c_float, c_uint32, c_void_p, c_void_p]),
    "anemoi_gt_edge_attention": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                         c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_int64, c_int64, c_int, c_int, c_void_p]),
    "anemoi_gather_add_act": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p,
                                      c_void_p, c_int64, c_int64, c_int, c_int, c_void_p]),
    "anemoi_segment_sum": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    "anemoi_segment_sum_cat": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64,
                                       c_int, c_void_p]),
    "anemoi_mhsa_workspace_bytes": (c_int64, [c_int, c_int, c_int, c_int, c_int]),
    "anemoi_mhsa": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                            c_int, c_float, c_uint32, c_void_p, c_int, c_int, c_void_p]),
    "anemoi_mhsa_backward_workspace_bytes": (c_int64, [c_int, c_int, c_int, c_int, c_int]),
    "anemoi_mhsa_backward": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p,
                                     c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_uint32,
                                     c_void_p, c_int, c_int, c_void_p]),
    "anemoi_assemble_nodes": (c_int, [c_int, c_void_p, c_int, c_int, c_int, c_int64, c_int, c_void_p, c_int,
                                      c_void_p, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "anemoi_finalize_output": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int64, c_int, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "anemoi_assemble_node_rows": (c_int, [c_int, c_void_p, c_int, c_int64, c_int, c_void_p, c_int, c_void_p, c_int,
                                          c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "anemoi_finalize_output_rows": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_int64,
                                            c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "anemoi_bound_output": (c_int, [c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                    c_void_p, c_void_p, c_void_p, c_void_p]),
    "anemoi_prognostic_residual": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int64, c_int, c_void_p,
                                           c_void_p, c_int, c_void_p]),
    "anemoi_advance_input": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_int, c_void_p, c_int, c_void_p, c_int,
                                     c_void_p, c_void_p]),
    "anemoi_transpose": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    "anemoi_transpose_colsum_rows": (c_int64, [c_int64, c_int64]),
    "anemoi_transpose_chunked": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int64, c_void_p,
                                         c_void_p]),
    "anemoi_weight_grad_tn": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64,
                                      c_int, c_int, c_int, c_void_p]),
    "anemoi_linear_batched": (c_int, [c_int, c_int, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                      c_int64, c_int, c_int64, c_int, c_int, c_void_p]),
    "anemoi_col_sum_workspace_floats": (c_int64, [c_int64, c_int]),
    "anemoi_col_sum": (c_int, [c_int, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_int64, c_void_p]),
    "anemoi_row_dot": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "anemoi_row_scale": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_float, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    "anemoi_act_forward": (c_int, [c_int, c_int, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int,
                                   c_void_p]),
    "anemoi_act_backward": (c_int, [c_int, c_int, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int,
                                    c_void_p]),
    "anemoi_layer_norm_backward_workspace_floats": (c_int64, [c_int64, c_int]),
    "anemoi_layer_norm_backward": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                                           c_int64, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                           c_int64, c_void_p]),
    "anemoi_gt_edge_attention_folded_backward_dst": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64,
                                                             c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                                             c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                                             c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                                             c_void_p, c_int64, c_int64, c_int, c_int, c_void_p]),
    "anemoi_gt_edge_attention_folded_backward_src": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_int64, c_void_p,
                                                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                             c_void_p, c_int64, c_int64, c_int, c_int, c_void_p]),
    "anemoi_gt_edge_attr_grad": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                         c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "anemoi_convert_pad": (c_int, [c_int, c_void_p, c_int64, c_int, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    "anemoi_add": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    "anemoi_gt_block_tail": (c_int, [ctypes.POINTER(GtBlockArgs), c_void_p]),
    "anemoi_gt_processor_block_forward": (c_int, [ctypes.POINTER(GtBlockArgs), c_void_p]),
    "anemoi_transformer_block_forward": (c_int, [ctypes.POINTER(TfmBlockArgs), c_void_p]),
}

_lib = None


class KernelLibraryMissing(RuntimeError):
    pass


def load(build_if_missing: bool = False) -> ctypes.CDLL:
    """Load the kernel library (once).  Raises :class:`KernelLibraryMissing` if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        if build_if_missing:
            from ._build import build

            build()
        else:
            raise KernelLibraryMissing(
                f"{LIB_PATH} not found: the HIP kernel library has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback."
            )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = ABI mismatch between header and library
        fn.restype = restype
        fn.argtypes = argtypes
    got = lib.anemoi_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"libanemoi_amd.so ABI version {got} != expected {ABI_VERSION}; rebuild the library")
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    """Map a C status code to the exception type the reference raises in the same situation."""
    if status == ANEMOI_OK:
        return
    msg = load().anemoi_last_error()
    msg = msg.decode() if msg else what
    if status == ANEMOI_ERR_INVALID:
        raise ValueError(msg)
    if status == ANEMOI_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise RuntimeError(msg)
