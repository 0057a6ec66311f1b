"""ctypes binding of the lab library (tools/lab_edge_mfma/build.sh)."""
import ctypes
import os
from ctypes import c_int, c_int64, c_void_p

import torch

from anemoi_models_amd import ops

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "liblab_edge_mfma.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise RuntimeError(f"{_PATH} missing: run tools/lab_edge_mfma/build.sh")
        _lib = ctypes.CDLL(_PATH)
        _lib.lab_gt_edge_attention_tiles.restype = c_int
        _lib.lab_gt_edge_attention_tiles.argtypes = [
            c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_void_p,
            c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64,
            c_int, c_int, c_void_p]
        _lib.lab_last_error.restype = ctypes.c_char_p
    return _lib


def gt_edge_attention_tiles(q, k, v, x_r, u, edge_attr, rowptr, tiles, num_heads, up, out=None, ld_out=None, lse=None):
    """Same arguments and result as ops.gt_edge_attention_folded with the tiling in place of the CSR column array."""
    n_dst, c = q.shape
    width = c + num_heads * up
    ld = width if ld_out is None else ld_out
    if out is None:
        out = torch.empty((n_dst, ld), dtype=q.dtype, device=q.device)
        if ld > width:
            out[:, width:].zero_()
    st = lib().lab_gt_edge_attention_tiles(
        q.data_ptr(), q.stride(0), k.data_ptr(), v.data_ptr(), k.stride(0), None if x_r is None else x_r.data_ptr(),
        0 if x_r is None else x_r.stride(0), u.data_ptr(), u.stride(0), edge_attr.data_ptr(), up, rowptr.data_ptr(),
        tiles.tiles.data_ptr(), tiles.tile_src.data_ptr(), tiles.posdst.data_ptr(), tiles.n_tiles, tiles.src_cap,
        tiles.edge_cap, out.data_ptr(), out.stride(0), None if lse is None else lse.data_ptr(), n_dst, k.shape[0],
        edge_attr.shape[0], c, num_heads, torch.cuda.current_stream().cuda_stream)
    if st != 0:
        raise RuntimeError(lib().lab_last_error().decode())
    return out
