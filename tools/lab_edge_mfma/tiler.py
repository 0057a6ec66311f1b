"""Destination tiling for the lab MFMA tile edge kernel (tools/lab_edge_mfma/edge_attention_mfma.hip).  Lab code: not part of
the product package."""
from dataclasses import dataclass
from typing import Optional

import torch
from torch import Tensor

from anemoi_models_amd.runtime import EdgePlan

# ---- destination tiles of the MFMA edge kernel (csrc/edge_attention_mfma.hip): 16 destinations per tile, and per
#      instantiation (sources, edges) capacities
MFMA_TILE_DST = 16
MFMA_TILE_CAPS = ((32, 64), (64, 160))


@dataclass
class EdgeMfmaTiles:
    """Tiling of a destination-sorted plan for ``anemoi_gt_edge_attention_tiles``."""

    tiles: Tensor  # int32 [n_tiles, 8]: first destination, destinations, offset into tile_src, sources, first edge, edges, 0, 0
    tile_src: Tensor  # int32: per tile its distinct source rows, ascending
    posdst: Tensor  # int32 (bit pattern of uint32) [E]: dense-image position | local destination << 16
    n_tiles: int
    src_cap: int  # capacities the tiling was built for (they select the kernel instantiation)
    edge_cap: int
    reuse: float  # edges per staged (tile, source) row: what one staged row replaces in gathers
    fill: float  # mean destinations per tile / 16: occupancy of the MFMA column block


def edge_mfma_tiles(plan: EdgePlan, src_cap: int = 64, edge_cap: int = 160) -> Optional[EdgeMfmaTiles]:
    """Tiles of <= 16 consecutive destinations with <= ``src_cap`` distinct sources and <= ``edge_cap`` edges, built once
    per plan with torch ops on the plan's device: start from aligned 16-row tiles and halve every tile that breaks a cap
    (16 -> 8 -> ... -> 1; aligned halves keep the tiles consecutive destination ranges).  ``None`` when a single
    destination breaks a cap: the caller keeps the gather kernel.  Integer outputs, checked against a scalar restatement
    in tests/test_host_logic.py."""
    cache = plan.__dict__.setdefault("_mfma_tiles", {})
    if (src_cap, edge_cap) in cache:
        return cache[(src_cap, edge_cap)]
    out = _build_mfma_tiles(plan, src_cap, edge_cap)
    cache[(src_cap, edge_cap)] = out
    return out


def _build_mfma_tiles(plan: EdgePlan, src_cap: int, edge_cap: int) -> Optional[EdgeMfmaTiles]:
    dev, n_dst, n_src, e = plan.rowptr.device, plan.n_dst, plan.n_src, plan.num_edges
    if e == 0 or n_dst == 0:
        return None
    dst_of_edge, col = plan.dst.long(), plan.col.long()
    d_idx = torch.arange(n_dst, device=dev)
    key = (d_idx // MFMA_TILE_DST) * MFMA_TILE_DST  # tile = the aligned range that starts at `key`
    size = MFMA_TILE_DST
    while True:
        ekey = key[dst_of_edge]
        uniq = torch.unique(ekey * n_src + col)
        n_s = torch.zeros(n_dst, dtype=torch.int64, device=dev).index_add_(0, uniq // n_src, torch.ones_like(uniq))
        n_e = torch.zeros(n_dst, dtype=torch.int64, device=dev).index_add_(0, ekey, torch.ones_like(ekey))
        bad = (n_s > src_cap) | (n_e > edge_cap)  # indexed by tile key (= its first destination)
        if not bool(bad.any()):
            break
        if size == 1:
            return None
        size //= 2
        split = bad[key]  # destinations of a tile that breaks a cap: re-keyed to the half-size aligned range
        key = torch.where(split, (d_idx // size) * size, key)
    starts = torch.unique(key)  # ascending first destinations
    n_tiles = starts.numel()
    tile_of_dst = torch.searchsorted(starts, key)
    nd = torch.bincount(tile_of_dst, minlength=n_tiles)
    etile = tile_of_dst[dst_of_edge]
    uniq, inverse = torch.unique(etile * n_src + col, return_inverse=True)  # tile-major, sources ascending
    ns = torch.bincount(uniq // n_src, minlength=n_tiles)
    s0 = torch.cumsum(ns, 0) - ns
    sl = inverse - s0[etile]  # slot of the edge's source in its tile's list
    dl = dst_of_edge - starts[etile]
    pos = ((sl >> 4) * 4 + ((sl >> 2) & 3)) * 64 + dl * 4 + (sl & 3)
    posdst = (pos | (dl << 16)).to(torch.int32).contiguous()
    e0 = plan.rowptr.long()[starts]
    ne = plan.rowptr.long()[starts + nd] - e0
    zero = torch.zeros_like(starts)
    tiles = torch.stack([starts, nd, s0, ns, e0, ne, zero, zero], dim=1).to(torch.int32).contiguous()
    return EdgeMfmaTiles(tiles, (uniq % n_src).to(torch.int32).contiguous(), posdst, int(n_tiles), src_cap, edge_cap,
                         float(e) / float(uniq.numel()), float(n_dst) / float(n_tiles) / MFMA_TILE_DST)


def use_edge_mfma_tiles(plan: EdgePlan, dtype: torch.dtype, channels: int, num_heads: int, up: int) -> Optional[EdgeMfmaTiles]:
    """The tiling when the MFMA tile kernel should run this edge set (bf16, heads of 64 channels, a multiple of 4 heads,
    both operand matrices below 2 GiB), else ``None`` -> the gather kernel.  It pays where a staged source row replaces
    several gathers: the processor mesh (2.3 edges per staged row) and the decoder (3 edges per grid node out of a handful
    of mesh rows per tile); not the encoder (21 edges per mesh node from 1.1 x as many distinct grid rows).  The small
    capacities are tried first (half the LDS: three workgroups per CU)."""
    if (dtype != torch.bfloat16 or channels != 64 * num_heads or num_heads % 4 != 0 or num_heads * up > 256
            or up not in (4, 8, 12, 16) or plan.num_edges == 0):
        return None
    for src_cap, edge_cap in MFMA_TILE_CAPS:
        tiles = edge_mfma_tiles(plan, src_cap, edge_cap)
        if tiles is not None and tiles.reuse >= 1.6 and tiles.fill >= 0.85:
            return tiles
    return None


