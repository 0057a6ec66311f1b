"""Host-side runtime state shared by the layer mirrors: compute dtype policy, dst-sorted CSR plans
of the edge sets, and packed (concatenated / cast / K-padded) weights.

Plans and packed weights are plumbing built with torch ops on the tensors' own device and cached;
they are recomputed when the underlying tensor is replaced or modified in place (``_version``).
"""

from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Callable, Optional, Sequence

import torch
from torch import Tensor

from . import ops

_FORCED_DTYPE = {"fp32": torch.float32, "float32": torch.float32, "bf16": torch.bfloat16,
                 "bfloat16": torch.bfloat16}


def compute_dtype(x: Tensor) -> torch.dtype:
    """Storage type of activations / GEMM weights for this call.

    ``ANEMOI_AMD_DTYPE=fp32|bf16`` forces it; otherwise an active CUDA autocast context selects its
    dtype (what anemoi-training's ``precision: 16-mixed / bf16-mixed`` does to the reference) and
    without autocast the input dtype is kept (f32 in -> exact-f32 kernels).
    """
    forced = os.environ.get("ANEMOI_AMD_DTYPE")
    if forced:
        return _FORCED_DTYPE[forced.lower()]
    if torch.is_autocast_enabled():
        dt = torch.get_autocast_dtype("cuda")
        if dt == torch.float16:
            # anemoi-training's ``precision: 16-mixed``: the reference's AutocastLayerNorm exists for "(b)float16" mixed
            # precision (layers/utils.py:33-39).  gfx950's MFMA rate is the same for fp16 and bf16 and this package has ONE
            # 16-bit kernel family, bf16 (8 exponent bits: no loss scaling needed, the f32 range of the residual stream
            # survives): a float16 autocast region runs on it.  Said once, loudly -- the results differ from fp16 arithmetic
            # in the last bits of the 16-bit roundings, the f32 accumulation is the same.
            global _FP16_NOTICE
            if not _FP16_NOTICE:
                _FP16_NOTICE = True
                import warnings

                warnings.warn("anemoi_models_amd: torch.autocast(float16) runs on the bf16 kernels of the MI355X path "
                              "(same 16-bit storage and MFMA rate, f32 accumulation; no fp16 kernels exist)", stacklevel=2)
            return torch.bfloat16
        if dt not in (torch.bfloat16, torch.float32):
            raise NotImplementedError(f"autocast dtype {dt} is not supported on the MI355X path (use bfloat16)")
        return dt
    return torch.bfloat16 if x.dtype == torch.bfloat16 else torch.float32


_FP16_NOTICE = False


def require_inference(*modules: torch.nn.Module) -> None:
    """Guard of the inference-only entry points (``native`` routes on packed weights): refuse to run where autograd would
    expect a graph.  The modules' ``forward`` methods take their differentiable routes (``training.py``) before they get here."""
    if torch.is_grad_enabled() and any(p.requires_grad for m in modules for p in m.parameters()):
        raise NotImplementedError(
            "anemoi_models_amd: this entry point is inference-only (packed weights, no autograd graph); call the module's "
            "forward, or run under torch.no_grad() / torch.inference_mode()"
        )


# ------------------------------------------------------------------------------------------ CSR plans
@dataclass
class EdgePlan:
    """Destination-sorted CSR view of an ``int64 [2, E]`` edge index (row 0 = src, row 1 = dst).

    ``perm[e]`` = original edge id at CSR slot ``e``.  The sort is stable, so the edges of one
    destination keep their original relative order (= the summation order of the reference's
    ``scatter_add_``).
    """

    rowptr: Tensor  # int32 [n_dst + 1]
    col: Tensor  # int32 [E]  source node of CSR slot e
    perm: Tensor  # int32 [E]
    n_src: int
    n_dst: int

    @property
    def num_edges(self) -> int:
        return int(self.col.shape[0])

    def runs3(self):
        """The shared-source lists of a uniform-degree-3 graph for the edge kernel (cached on the plan), or ``None`` for any
        other graph: the GROUPS ``(grp_ptr, grp_perm, grp_dst)`` of :func:`_groups3` (all destinations of a source triple,
        ``anemoi_gt_edge_attention_folded_groups``), or -- with ``ANEMOI_AMD_EDGE_GROUPS=0`` -- the consecutive RUNS
        ``(run_ptr, perm)`` of :func:`_runs3`."""
        r = getattr(self, "_runs3_cache", 0)
        if r == 0:
            r = None
            if self.col.is_cuda:
                if os.environ.get("ANEMOI_AMD_EDGE_GROUPS", "1") != "0":
                    r = _groups3(self)
                if r is None:
                    r = _runs3(self)
            self._runs3_cache = r
        return r

    def schedule(self, dtype: torch.dtype, channels: int):
        """Destination schedule of ``anemoi_gt_edge_attention_folded_sched`` for this plan at ``channels`` / ``dtype``
        (cached on the plan): int32 ``[8, slots, steps]`` on the plan's device, or ``None`` where the scheduled kernel has
        nothing to offer (f32; graphs of uniform in-degree 3, which take the run kernel)."""
        if dtype != torch.bfloat16 or not self.col.is_cuda or self.n_dst == 0:
            return None
        cache = self.__dict__.setdefault("_sched_cache", {})
        key = (dtype, channels)
        if key not in cache:
            cache[key] = _edge_schedule(self, dtype, channels)
        return cache[key]

    def tiles(self, dtype: torch.dtype, channels: int, heads: int, up: int):
        """:class:`EdgeTiles` of this plan for the LDS-tile edge kernel (``anemoi_gt_edge_attention_folded_tiles``; cached on
        the plan), or ``None`` where that kernel does not apply."""
        if dtype != torch.bfloat16 or not self.col.is_cuda:
            return None
        cache = self.__dict__.setdefault("_tiles_cache", {})
        key = (channels, heads, up)
        if key not in cache:
            cache[key] = _edge_tiles(self, channels, heads, up)
        return cache[key]

    @property
    def dst(self) -> Tensor:
        """int32 [E]: destination node of every CSR slot (row index expanded; built on first use)."""
        d = getattr(self, "_dst", None)
        if d is None:
            counts = (self.rowptr[1:] - self.rowptr[:-1]).long()
            d = torch.repeat_interleave(torch.arange(self.n_dst, device=self.rowptr.device), counts).to(torch.int32)
            self._dst = d
        return d


# relative cost of a destination for the schedule: a fixed part (q / u / x_r / out streams, epilogue) + one unit per chunk of
# SCHED_U in-edges (the kernel's gather batch)
SCHED_U, SCHED_UNIT_COST = 4, 1.5
# graphs of a higher mean in-degree stay on the round-robin kernel: the N320 -> ico-6 encoder (mean 21, 16 ... 104) measured
# 0.588 ms there, 0.585 with an identity schedule and 0.630 with the balanced one (the prefetched ids cover 8 of its edges, and
# re-dealing its destinations costs more L2 locality than the balance returns)
SCHED_MAX_MEAN_DEGREE = 12


def edge_schedule_lists(degree: Tensor, slots: int, steps: int, chunk: int = SCHED_U, unit_cost: float = SCHED_UNIT_COST) -> Tensor:
    """The static destination schedule of the scheduled edge kernel (host logic, CPU tensors): int32 ``[8, slots, steps]``.

    XCD ``x`` owns the destinations ``[n x / 8, n (x + 1) / 8)``; at step ``i`` its ``slots`` wave slots take the ``i``-th
    group of ``slots`` consecutive destinations (so that what is in flight at any time is one contiguous window of the
    CSR, as in the round-robin kernel), the most expensive destination of the group going to the slot with the least work so
    far.  Cost of a destination = ``unit_cost + ceil(degree / chunk)``.  Unused entries are -1; every list ends with >= 3."""
    n = int(degree.shape[0])
    cost = unit_cost + torch.div(degree.to(torch.float64) + (chunk - 1), chunk, rounding_mode="floor")
    sched = torch.full((8, slots, steps), -1, dtype=torch.int32)
    uniform = n == 0 or bool((degree == degree[0]).all()) or os.environ.get("ANEMOI_AMD_EDGE_BALANCE", "1") == "0"  # (lab: A/B)
    for x in range(8):
        n0, n1 = n * x // 8, n * (x + 1) // 8
        load = torch.zeros(slots, dtype=torch.float64)
        for i, g0 in enumerate(range(n0, n1, slots)):
            ids = torch.arange(g0, min(g0 + slots, n1), dtype=torch.int32)
            k = ids.shape[0]
            if uniform:
                sched[x, :k, i] = ids
                continue
            c = cost[g0:g0 + k]
            by_cost = torch.argsort(c, descending=True, stable=True)
            by_load = torch.argsort(load, stable=True)[:k]
            sched[x, by_load, i] = ids[by_cost]
            load[by_load] += c[by_cost]
    return sched


def _edge_schedule(plan: "EdgePlan", dtype: torch.dtype, channels: int):
    import ctypes

    from . import _lib

    slots, steps = ctypes.c_int(0), ctypes.c_int(0)
    st = _lib.load().anemoi_edge_schedule_shape(ops.dtype_code(dtype), plan.n_dst, channels, ctypes.byref(slots),
                                                ctypes.byref(steps))
    _lib.check(st, "anemoi_edge_schedule_shape")
    degree = (plan.rowptr[1:] - plan.rowptr[:-1]).cpu()
    if plan.num_edges > SCHED_MAX_MEAN_DEGREE * plan.n_dst:
        return None
    return edge_schedule_lists(degree, slots.value, steps.value).to(plan.col.device).contiguous()


# ---- destination TILES whose source rows are staged in LDS (round 6: anemoi_gt_edge_attention_folded_tiles) ------------
TILE_MAX_DST = 32        # destinations per tile: 4 waves x 2 passes x 4 destinations (one per 16-lane row of a wave)
TILE_SLICE = 128         # channels per workgroup: 16 lanes x 8 channels per destination, 256 bytes of a bf16 k (v) row
TILE_SRC_CAP = 72        # distinct sources a tile stages: 72 x 512 bytes of k|v slices ...
TILE_EDGE_CAP = 256      # ... next to the tile's attribute rows (256 x up x 4 bytes) and slot bytes: 3 workgroups per CU


@dataclass
class EdgeTiles:
    """Host-built lists of the LDS-tile edge kernel (``anemoi_gt_edge_attention_folded_tiles``; int32 / uint8, on the plan's device).

    A TILE is a run of <= ``TILE_MAX_DST`` consecutive destinations (Morton order: neighbours in space) of one XCD's
    range whose in-edges name <= ``src_cap`` distinct sources and <= ``edge_cap`` edges.  The kernel stages the k|v slices of
    the tile's sources ONCE in LDS; every edge then reads its source by LDS slot instead of gathering the row again -- on the
    ico-6 multi-scale mesh a staged row serves 2.8 edges.
      hdr   [n_tiles, 8]        first CSR slot of the tile, its edge count, offset of its source list, source count, offset of
                                its slot bytes (a multiple of 16), destination count, 0, 0
      dst   [n_tiles, 32, 2]    per (pass, lane row): destination id (-1: none), (first edge - tile's first edge) << 8 | degree;
                                pass p = the p-th four of the tile's destinations by descending in-degree (rows of one pass
                                run in lockstep: similar degrees waste the fewest lanes; heavy passes are started first)
      src   [sum of counts]     the tiles' distinct sources, ascending per tile
      slot  [...]               uint8 per tile and edge (CSR order inside the tile): index of the edge's source in the tile's list
      xcd   [9]                 tile index ranges of the 8 XCDs (XCD x: destinations [n x / 8, n (x + 1) / 8))"""

    hdr: Tensor
    dst: Tensor
    src: Tensor
    slot: Tensor
    xcd: Tensor
    src_cap: int
    edge_cap: int
    max_tiles_per_xcd: int

    @property
    def n_tiles(self) -> int:
        return int(self.hdr.shape[0])

    def to(self, device) -> "EdgeTiles":
        return EdgeTiles(self.hdr.to(device), self.dst.to(device), self.src.to(device), self.slot.to(device),
                         self.xcd.to(device), self.src_cap, self.edge_cap, self.max_tiles_per_xcd)


def edge_tile_lists(rowptr: Tensor, col: Tensor, src_cap: int = TILE_SRC_CAP, edge_cap: int = TILE_EDGE_CAP,
                    max_dst: int = TILE_MAX_DST):
    """The tile lists of :class:`EdgeTiles` for a destination-sorted CSR (host logic, CPU tensors), or ``None`` when one
    destination alone exceeds a cap (in-degree > ``edge_cap`` / 255, more than ``src_cap`` distinct sources)."""
    import numpy as np

    rp = rowptr.cpu().numpy().astype(np.int64)
    cl = col.cpu().numpy().astype(np.int64)
    n = rp.shape[0] - 1
    deg = rp[1:] - rp[:-1]
    if (n == 0 or not 0 < src_cap <= 255 or edge_cap % 16 != 0 or max_dst > TILE_MAX_DST
            or deg.max(initial=0) > min(edge_cap, 255)):
        return None
    hdr, dst, srcs, slots, xcd = [], [], [], [], [0]
    src_off = slot_off = 0
    for x in range(8):
        n0, n1 = n * x // 8, n * (x + 1) // 8
        d = n0
        while d < n1:
            first, cur, edges = d, set(), 0
            while d < n1 and d - first < max_dst:
                mine = set(cl[rp[d]:rp[d + 1]].tolist())
                if len(cur | mine) > src_cap or edges + int(deg[d]) > edge_cap:
                    break
                cur |= mine
                edges += int(deg[d])
                d += 1
            if d == first:
                return None  # (one destination alone does not fit)
            ids = np.array(sorted(cur), dtype=np.int64)
            e0, e1 = int(rp[first]), int(rp[d])
            sl = np.zeros((e1 - e0 + 15) // 16 * 16, dtype=np.uint8)
            sl[:e1 - e0] = np.searchsorted(ids, cl[e0:e1])
            hdr.append((e0, e1 - e0, src_off, ids.shape[0], slot_off, d - first, 0, 0))
            srcs.append(ids)
            slots.append(sl)
            src_off += ids.shape[0]
            slot_off += sl.shape[0]
            order = np.argsort(-deg[first:d], kind="stable") + first  # heavy destinations first
            info = np.zeros((TILE_MAX_DST, 2), dtype=np.int64)
            info[:, 0] = -1
            info[:order.shape[0], 0] = order
            info[:order.shape[0], 1] = ((rp[order] - e0) << 8) | deg[order]
            dst.append(info)
        xcd.append(len(hdr))
    as_i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a).astype(np.int32))  # noqa: E731
    per_xcd = max(xcd[i + 1] - xcd[i] for i in range(8))
    slot = np.concatenate(slots) if slots else np.zeros(16, dtype=np.uint8)
    return EdgeTiles(as_i32(np.array(hdr)), as_i32(np.stack(dst)), as_i32(np.concatenate(srcs)),
                     torch.from_numpy(np.concatenate([slot, np.zeros(16, dtype=np.uint8)])), as_i32(np.array(xcd)), src_cap,
                     edge_cap, per_xcd)


def _edge_tiles(plan: "EdgePlan", channels: int, heads: int, up: int):
    """:class:`EdgeTiles` of a plan on its device, or ``None`` where the tile kernel does not apply: channel counts off the
    128-channel slice, head sizes other than 32 / 64, graphs of a mean in-degree above ``SCHED_MAX_MEAN_DEGREE`` (the
    encoder: its tiles would hold 3 destinations) or uniform degree 3 (the decoder: the group kernel)."""
    if (channels % TILE_SLICE != 0 or heads <= 0 or channels % heads != 0 or channels // heads not in (32, 64)
            or up not in (4, 8, 12, 16) or plan.n_dst == 0 or plan.num_edges > SCHED_MAX_MEAN_DEGREE * plan.n_dst):
        return None
    src_cap = int(os.environ.get("ANEMOI_AMD_EDGE_TILE_SRC", TILE_SRC_CAP))      # lab switches (A/B of the LDS budget)
    edge_cap = int(os.environ.get("ANEMOI_AMD_EDGE_TILE_EDGES", TILE_EDGE_CAP))
    t = edge_tile_lists(plan.rowptr, plan.col, src_cap, edge_cap)
    return None if t is None else t.to(plan.col.device)


def _runs3(plan: "EdgePlan", max_run: int = 2):
    """Runs of consecutive destinations with the same three sources, for ``anemoi_gt_edge_attention_folded_runs``:
    ``(run_ptr int32 [n_runs + 1], perm int32 [n_runs])`` when every destination of ``plan`` has exactly three in-edges from
    three different sources (the reference's mesh -> grid decoder on anemoi-graphs' 3-nearest-neighbour edges), else
    ``None`` -- also when the runs are too short to pay (mean length below 1.4 at the cap of ``max_run`` = 2 the kernel is
    built for: at O96 -> ico-5, mean 1.22, the run kernel is 20 % SLOWER than the plain one).  ``perm[r]`` packs 6 bits per
    destination d = 0, 1 of run r: for s = 0 .. 2 the position inside d's CSR segment
    of its edge to the s-th source in ascending source order.  Built once per plan on the plan's device, no host round trip besides two scalar checks."""
    n, e = plan.n_dst, plan.num_edges
    if n < 1024 or e != 3 * n:
        return None
    rowptr = plan.rowptr
    if not bool((rowptr == torch.arange(0, 3 * n + 1, 3, dtype=rowptr.dtype, device=rowptr.device)).all()):
        return None
    src = plan.col.view(n, 3).long()
    srt, pos = torch.sort(src, dim=1, stable=True)
    if bool((srt[:, 1:] == srt[:, :-1]).any()):
        return None  # a destination with two edges from one source: the run kernel's canonical order would be ambiguous
    perm = pos[:, 0] | (pos[:, 1] << 2) | (pos[:, 2] << 4)  # per destination, 6 bits
    start = torch.ones(n, dtype=torch.bool, device=src.device)
    start[1:] = (srt[1:] != srt[:-1]).any(dim=1)
    # cap the run length: position inside its run, a new run every max_run destinations
    idx = torch.arange(n, device=src.device)
    first = torch.cummax(torch.where(start, idx, torch.zeros_like(idx)), 0).values
    start |= ((idx - first) % max_run) == 0
    begin = torch.nonzero(start).flatten()
    if n < 1.4 * begin.shape[0]:
        return None
    run_ptr = torch.cat([begin, torch.tensor([n], device=src.device)])
    # per run: the permutations of its (at most max_run = 4) destinations, 6 bits each
    lens = run_ptr[1:] - run_ptr[:-1]
    packed = torch.zeros_like(begin)
    for d in range(max_run):
        has = lens > d
        packed[has] |= perm[(begin + d)[has]] << (6 * d)
    return run_ptr.to(torch.int32).contiguous(), packed.to(torch.int32).contiguous()


def _groups3(plan: "EdgePlan", max_group: int = 8):
    """Groups of destinations with the same three sources, wherever they lie in the destination order, for
    ``anemoi_gt_edge_attention_folded_groups``: ``(grp_ptr int32 [n_groups + 1], grp_perm int32 [n_dst], grp_dst int32
    [n_dst])`` under the conditions of :func:`_runs3` (exactly three in-edges from three different sources per destination),
    else ``None`` -- also when the groups are too short to pay (mean below 1.5).  ``grp_dst`` lists every destination once,
    sorted by (ascending) source triple and, inside a triple, by destination; a triple's destinations are cut into groups of
    at most ``max_group`` (the kernel's ``EDGE_MAX_GROUP``).  ``grp_perm[i]``: 6 bits, for s = 0 .. 2 the position inside the
    CSR segment of ``grp_dst[i]`` of its edge to the s-th source in ascending source order.  Built on the plan's device."""
    n, e = plan.n_dst, plan.num_edges
    if n < 1024 or e != 3 * n or plan.n_src >= 2**21:  # (the triple is packed into one 63-bit key)
        return None
    rowptr = plan.rowptr
    if not bool((rowptr == torch.arange(0, 3 * n + 1, 3, dtype=rowptr.dtype, device=rowptr.device)).all()):
        return None
    src = plan.col.view(n, 3).long()
    srt, pos = torch.sort(src, dim=1, stable=True)
    if bool((srt[:, 1:] == srt[:, :-1]).any()):
        return None
    perm = pos[:, 0] | (pos[:, 1] << 2) | (pos[:, 2] << 4)
    key = (srt[:, 0] * plan.n_src + srt[:, 1]) * plan.n_src + srt[:, 2]
    order = torch.argsort(key, stable=True)
    sk = key[order]
    start = torch.ones(n, dtype=torch.bool, device=src.device)
    start[1:] = sk[1:] != sk[:-1]
    idx = torch.arange(n, device=src.device)
    first = torch.cummax(torch.where(start, idx, torch.zeros_like(idx)), 0).values
    start |= ((idx - first) % max_group) == 0
    begin = torch.nonzero(start).flatten()
    if n < 1.5 * begin.shape[0]:
        return None
    grp_ptr = torch.cat([begin, torch.tensor([n], device=src.device)])
    return (grp_ptr.to(torch.int32).contiguous(), perm[order].to(torch.int32).contiguous(),
            order.to(torch.int32).contiguous())


def build_edge_plan(edge_index: Tensor, n_src: int, n_dst: int) -> EdgePlan:
    if edge_index.dim() != 2 or edge_index.shape[0] != 2:
        raise ValueError(f"edge_index must have shape [2, E], got {tuple(edge_index.shape)}")
    e = edge_index.shape[1]
    if e >= 2**31 or n_src >= 2**31 or n_dst >= 2**31:
        raise NotImplementedError("edge / node counts beyond int32 are not supported")
    src, dst = edge_index[0], edge_index[1]
    if e > 0:
        smin, smax = int(src.min()), int(src.max())
        dmin, dmax = int(dst.min()), int(dst.max())
        if smin < 0 or smax >= n_src or dmin < 0 or dmax >= n_dst:
            raise ValueError(
                f"edge_index out of range: src in [{smin}, {smax}] for {n_src} source nodes, "
                f"dst in [{dmin}, {dmax}] for {n_dst} destination nodes"
            )
    perm = torch.argsort(dst, stable=True)
    counts = torch.bincount(dst, minlength=n_dst)
    rowptr = torch.zeros(n_dst + 1, dtype=torch.int64, device=edge_index.device)
    torch.cumsum(counts, 0, out=rowptr[1:])
    return EdgePlan(rowptr.to(torch.int32), src[perm].to(torch.int32).contiguous(), perm.to(torch.int32), n_src,
                    n_dst)


_PLAN_DIR: Optional[str] = None


def set_plan_cache_dir(path: Optional[str]) -> None:
    """Directory of the on-disk plan cache (``None`` = off unless ``ANEMOI_AMD_PLAN_CACHE_DIR`` is set).

    A plan is a pure function of the graph (edge index, node relabelling, batch size), so it is keyed on a content hash
    of exactly those inputs: the same anemoi-graphs file gives the same keys in every process, on every rank, across
    restarts.  Files are written atomically (temporary name + rename): concurrent ranks may race to
    write the same plan and all of them read a complete file."""
    global _PLAN_DIR
    _PLAN_DIR = path


def plan_cache_dir() -> Optional[str]:
    return _PLAN_DIR or os.environ.get("ANEMOI_AMD_PLAN_CACHE_DIR") or None


def tensor_digest(*items) -> str:
    """Content hash (BLAKE2b-128, hex) of tensors / ints / strings / None, order sensitive."""
    import hashlib

    h = hashlib.blake2b(digest_size=16)
    for it in items:
        if isinstance(it, Tensor):
            t = it.detach().cpu().contiguous()
            h.update(f"T{t.dtype}{tuple(t.shape)}".encode())
            h.update(t.view(torch.uint8).numpy().tobytes() if t.numel() else b"")
        else:
            h.update(f"V{it!r}".encode())
    return h.hexdigest()


def graph_hash(graph) -> str:
    """Content hash of a graph object (``HeteroData`` or this package's ``GraphData``): node coordinates, edge indices
    and edge attributes of every store, in a canonical order -- the identity of an anemoi-graphs file's CONTENT."""
    items = []
    for name, store in sorted(graph.node_items(), key=lambda kv: kv[0]):
        items += ["node", name, store.x]
    edge_types = sorted(graph.edge_types) if hasattr(graph, "edge_types") else []
    for key in edge_types:
        store = graph[key]
        items += ["edge", "/".join(key), store["edge_index"]]
        for k in sorted(k for k in store.keys() if k != "edge_index"):
            v = store[k]
            if isinstance(v, Tensor):
                items += [k, v]
    return tensor_digest(*items)


class PlanCache:
    """Edge plans keyed by the identity + version of the edge-index tensor they were built from; behind it, when a
    directory is configured (:func:`set_plan_cache_dir`), an on-disk cache keyed on the CONTENT of the plan's inputs."""

    def __init__(self) -> None:
        self._plans: dict = {}

    @staticmethod
    def _disk_path(edge_index, n_src, n_dst, batch_size, edge_inc, src_map, dst_map) -> Optional[str]:
        root = plan_cache_dir()
        if root is None:
            return None
        key = tensor_digest("edgeplan-v1", edge_index, n_src, n_dst, batch_size, edge_inc if batch_size > 1 else None,
                            src_map, dst_map)
        return os.path.join(root, f"edgeplan-{key}.pt")

    def get(self, edge_index: Tensor, n_src: int, n_dst: int, batch_size: int = 1,
            edge_inc: Optional[Tensor] = None, src_map: Optional[Tensor] = None,
            dst_map: Optional[Tensor] = None) -> EdgePlan:
        """``src_map`` / ``dst_map`` (int64, external node id -> internal row) relabel the node sets, e.g. to the
        locality-preserving internal mesh order of :func:`locality_order`; they cover one batch block."""
        key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape), str(edge_index.device), n_src,
               n_dst, batch_size, None if src_map is None else src_map.data_ptr(),
               None if dst_map is None else dst_map.data_ptr())
        plan = self._plans.get(key)
        if plan is None:
            path = self._disk_path(edge_index, n_src, n_dst, batch_size, edge_inc, src_map, dst_map)
            if path is not None and os.path.exists(path):
                plan = load_edge_plan(path, edge_index.device, n_src, n_dst)
            if plan is None:
                ei = edge_index
                if src_map is not None or dst_map is not None:
                    src = ei[0] if src_map is None else src_map[ei[0]]
                    dst = ei[1] if dst_map is None else dst_map[ei[1]]
                    ei = torch.stack([src, dst])
                if batch_size > 1:
                    ei = expand_edges(ei, edge_inc, batch_size)
                plan = build_edge_plan(ei, n_src, n_dst)
                if path is not None:
                    save_edge_plan(plan, path)
            if len(self._plans) > 16:
                self._plans.clear()
            self._plans[key] = plan
        return plan


def save_edge_plan(plan: EdgePlan, path: str) -> None:
    os.makedirs(os.path.dirname(path), exist_ok=True)
    tmp = f"{path}.{os.getpid()}.tmp"
    torch.save({"format": "anemoi_models_amd.EdgePlan/1", "rowptr": plan.rowptr.cpu(), "col": plan.col.cpu(),
                "perm": plan.perm.cpu(), "n_src": plan.n_src, "n_dst": plan.n_dst}, tmp)
    os.replace(tmp, path)


def load_edge_plan(path: str, device, n_src: Optional[int] = None, n_dst: Optional[int] = None) -> Optional[EdgePlan]:
    """The plan stored at ``path`` on ``device``; ``None`` for a file that is not a complete, consistent plan for
    ``n_src`` x ``n_dst`` nodes (it is then rebuilt and overwritten).  A loaded plan is held to what a built one
    guarantees -- the edge kernels gather ``col`` rows and ``perm`` attribute rows without further checks: monotone row
    pointers ending at E, every column inside the source set, ``perm`` a permutation of the edges."""
    try:
        d = torch.load(path, map_location="cpu", weights_only=True)
        rowptr, col, perm = d["rowptr"], d["col"], d["perm"]
        n_edges = col.shape[0]
        ok = (d.get("format") == "anemoi_models_amd.EdgePlan/1" and rowptr.dtype == torch.int32
              and col.dtype == torch.int32 and perm.dtype == torch.int32
              and rowptr.dim() == 1 and col.dim() == 1 and rowptr.shape[0] == d["n_dst"] + 1 and col.shape == perm.shape
              and (n_src is None or int(d["n_src"]) == int(n_src)) and (n_dst is None or int(d["n_dst"]) == int(n_dst))
              and int(rowptr[0]) == 0 and int(rowptr[-1]) == n_edges and bool((rowptr[1:] >= rowptr[:-1]).all()))
        if ok and n_edges > 0:
            ok = (int(col.min()) >= 0 and int(col.max()) < int(d["n_src"])
                  and bool((torch.bincount(perm.long().clamp_(0, n_edges - 1), minlength=n_edges) == 1).all())
                  and int(perm.min()) >= 0 and int(perm.max()) < n_edges)
    except Exception:  # noqa: BLE001  (truncated / foreign file: fall back to building)
        return None
    if not ok:
        return None
    return EdgePlan(rowptr.to(device), col.to(device), perm.to(device), int(d["n_src"]), int(d["n_dst"]))


def locality_order(sincos_latlon: Tensor) -> Tensor:
    """Permutation (internal row -> external node id) that sorts nodes along a 3-D Morton curve of their positions.

    ``sincos_latlon`` is the ``[N, 4]`` buffer ``[sin lat, sin lon, cos lat, cos lon]`` the model keeps for every
    node set (reference layers/graph.py:90-93).  Refined icosahedral meshes number their nodes level by level, so
    graph neighbours are far apart in memory; along the Morton curve the in-neighbours of consecutive destination
    rows fall into a compact window, which is what lets the edge kernel's k/v gathers hit in the XCD-local L2.
    Purely internal: the mesh never appears in the model's inputs or outputs.
    """
    s_lat, s_lon, c_lat, c_lon = sincos_latlon.detach().double().unbind(dim=1)
    xyz = torch.stack([c_lat * c_lon, c_lat * s_lon, s_lat], dim=1)
    q = ((xyz + 1.0) * 0.5 * 1023.0).round().clamp_(0, 1023).to(torch.int64)
    code = torch.zeros(q.shape[0], dtype=torch.int64, device=q.device)
    for bit in range(10):
        for axis in range(3):
            code |= ((q[:, axis] >> bit) & 1) << (3 * bit + axis)
    return torch.argsort(code, stable=True)


def inverse_permutation(order: Tensor) -> Tensor:
    inv = torch.empty_like(order)
    inv[order] = torch.arange(order.shape[0], device=order.device, dtype=order.dtype)
    return inv


def expand_edges(edge_index: Tensor, edge_inc: Tensor, batch_size: int) -> Tensor:
    """Batched edge index: block ``i`` is ``edge_index + i * edge_inc`` (reference layers/mapper.py:150-171)."""
    return torch.cat([edge_index + i * edge_inc for i in range(batch_size)], dim=1)


# ------------------------------------------------------------------------------------------ packed weights
class PackedWeights:
    """Cache of derived weight tensors (concatenated, cast to the compute dtype, K padded)."""

    def __init__(self) -> None:
        self._store: dict = {}

    def get(self, key, params: Sequence[Optional[Tensor]], build: Callable[[], Tensor]) -> Tensor:
        ver = tuple((p.data_ptr(), p._version, p.device) for p in params if p is not None)
        hit = self._store.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
        with torch.no_grad():
            val = build()
        self._store[key] = (ver, val)
        return val

    def clear(self) -> None:
        self._store.clear()


def edge_attr_csr_cached(cache: PackedWeights, edge_attr: Tensor, trainable: Optional[Tensor], plan, *layout) -> Tensor:
    """``ops.edge_attr_csr(edge_attr, trainable, plan.perm, *layout)`` -- the sub-graph's edge attributes next to the
    trainable edge tensor, gathered into the plan's CSR order -- kept until one of the two tensors changes (pointer or
    in-place version, as every derived weight): in inference they never do, and the gather (1.6 M decoder edges at
    config 3) leaves the per-step stream.  The entry holds the plan, so its identity cannot be recycled."""
    def build():
        return plan, ops.edge_attr_csr(edge_attr, trainable, plan.perm, *layout)

    return cache.get(("edge_attr_csr", id(plan), layout), [edge_attr, trainable], build)[1]


def pack_weight(weights: Sequence[Tensor], dtype: torch.dtype, k_pad: Optional[int] = None) -> Tensor:
    """``cat(weights, 0)`` as ``[N, K_pad]`` in ``dtype`` with K zero padded to the kernel's K-slab multiple (or to
    ``k_pad`` columns when the activation carries more padding than that)."""
    w = torch.cat([t.detach() for t in weights], dim=0) if len(weights) > 1 else weights[0].detach()
    k = w.shape[1]
    kp = ops.round_up(k, ops.k_multiple(dtype)) if k_pad is None else k_pad
    out = torch.zeros((w.shape[0], kp), dtype=dtype, device=w.device)
    out[:, :k] = w.to(dtype)
    return out


def pack_weight_cols(parts: Sequence[Tensor], dtype: torch.dtype) -> Tensor:
    """``cat(parts, 1)`` (extra INPUT columns) as ``[N, K_pad]`` in ``dtype``, K zero padded to the K-slab multiple."""
    w = torch.cat([t.detach() for t in parts], dim=1)
    k = w.shape[1]
    kp = ops.round_up(k, ops.k_multiple(dtype))
    out = torch.zeros((w.shape[0], kp), dtype=dtype, device=w.device)
    out[:, :k] = w.to(dtype)
    return out


def ln_fold_enabled(dtype: torch.dtype) -> bool:
    """LayerNorm -> Linear pairs run as row_stats + anemoi_linear_ln (bf16 only; ``ANEMOI_AMD_LN_FOLD=0`` disables)."""
    return dtype == torch.bfloat16 and os.environ.get("ANEMOI_AMD_LN_FOLD", "1") != "0"


def fold_layer_norm(w_rows: Tensor, bias: Optional[Tensor], gamma: Tensor, beta: Tensor, dtype: torch.dtype):
    """Fold ``LayerNorm(gamma, beta)`` into the Linear ``(w_rows [N, K] f32, bias [N] f32 or None)`` that consumes it.

    Returns ``(W' [N, K_pad] in dtype, b' [N] f32, colsum [N] f32)`` with ``W' = W * gamma`` (input columns scaled),
    ``colsum = W'.sum(1)`` taken from the ROUNDED W' (so that the mean cancels exactly in the kernel's
    ``rstd (x W'^T) - rstd mean colsum``) and ``b' = b + W beta``.
    """
    w = w_rows.detach().float()
    g, bt = gamma.detach().float(), beta.detach().float()
    k = w.shape[1]
    kp = ops.round_up(k, ops.k_multiple(dtype))
    wp = torch.zeros((w.shape[0], kp), dtype=dtype, device=w.device)
    wp[:, :k] = (w * g[None, :]).to(dtype)
    colsum = wp.float().sum(dim=1).contiguous()
    b = w @ bt
    if bias is not None:
        b = b + bias.detach().float()
    return wp, b.contiguous(), colsum


def embed_fold_enabled(dtype: torch.dtype) -> bool:
    """Embedding -> LayerNorm -> Linear chains of the mappers run as ONE narrow GEMM on the raw node features
    (``fold_embedded_layer_norm``; bf16 LayerNorm-fold path only; ``ANEMOI_AMD_EMBED_FOLD=0`` disables)."""
    return ln_fold_enabled(dtype) and os.environ.get("ANEMOI_AMD_EMBED_FOLD", "1") != "0"


def _centered_embedding(emb_w: Tensor, emb_b: Optional[Tensor]):
    """``(E_c [C, K], b_c [C])`` in f64 with the mean over the C output channels removed: for h = E x + b,
    ``h - mean_c(h) = E_c x + b_c`` exactly -- the LayerNorm's mean subtraction moved into the parameters."""
    e = emb_w.detach().double()
    b = torch.zeros(e.shape[0], dtype=torch.float64, device=e.device) if emb_b is None else emb_b.detach().double()
    return e - e.mean(dim=0, keepdim=True), b - b.mean()


def fold_embedded_layer_norm(w_rows: Tensor, bias: Optional[Tensor], gamma: Tensor, beta: Tensor, emb_w: Tensor,
                             emb_b: Optional[Tensor], k_pad: int, one_col: int, dtype: torch.dtype):
    """Fold ``Linear(LayerNorm(emb(x)))`` (mapper embedding -> block LayerNorm -> q / k / v Linear, reference
    layers/mapper.py:322-331 + layers/block.py:516-528) into ONE product on the raw features ``x_aug = [x | 1 | 0-pad]``:

        Linear(LN(E x + b_e)) = rstd * ( F x_aug ) + b',   F = [ W' E_c | W' b_c ],  W' = W * gamma,  b' = b + W beta,

    with ``E_c, b_c`` the channel-centred embedding (``_centered_embedding``) and ``rstd`` the LayerNorm's own row
    statistic of ``E x + b_e``.  Nothing is subtracted in the epilogue (the mean is gone algebraically), so the
    ``colsum`` the kernel multiplies ``-mean rstd`` with is returned as zeros.  ``(F [N, k_pad] dtype, b' [N] f32,
    zeros [N] f32)``; products in f64, one rounding to ``dtype``."""
    w = w_rows.detach().double() * gamma.detach().double()[None, :]
    e_c, b_c = _centered_embedding(emb_w, emb_b)
    k_in = e_c.shape[1]
    if not (k_in <= one_col < k_pad):
        raise ValueError(f"fold_embedded_layer_norm: constant-1 column {one_col} outside the padding [{k_in}, {k_pad})")
    f = torch.zeros((w.shape[0], k_pad), dtype=torch.float64, device=w.device)
    f[:, :k_in] = w @ e_c
    f[:, one_col] = w @ b_c
    b = w_rows.detach().double() @ beta.detach().double()
    if bias is not None:
        b = b + bias.detach().double()
    return (f.to(dtype).contiguous(), b.float().contiguous(),
            torch.zeros(w.shape[0], dtype=torch.float32, device=w.device))


def embedding_stats_operator(emb_w: Tensor, emb_b: Optional[Tensor], k_pad: int, one_col: int, dtype: torch.dtype):
    """``T [n, k_pad]`` (n = a multiple of 256) such that ``y = T x_aug`` has, per row, mean 0 and
    ``sum(y^2) / n == |E_c x + b_c|^2 / C``: the LayerNorm variance of the embedded row ``E x + b_e`` without forming it.
    A plain ``row_stats(y, eps)`` (or the statistics epilogue of the GEMM that produces ``y``) then IS
    ``{rstd, ~0}`` of ``LayerNorm(emb(x))``.  Construction: ``A = [E_c | b_c] = Q R`` (so ``|A z| = |R z|``),
    ``B`` an orthonormal basis of ``K + 1`` directions orthogonal to the all-ones vector in ``R^n``,
    ``T = sqrt(n / C) B R`` -- sums of squares only, no inverse, any rank."""
    e_c, b_c = _centered_embedding(emb_w, emb_b)
    c, k_in = e_c.shape
    a = torch.cat([e_c, b_c[:, None]], dim=1).cpu()  # [C, K + 1]; small, factorised on the host in f64
    r = torch.linalg.qr(a, mode="r").R if c >= k_in + 1 else a
    n = ops.round_up(r.shape[0] + 1, 256)
    basis = torch.eye(n, dtype=torch.float64)[:, : r.shape[0]] - 1.0 / n
    basis = torch.linalg.qr(basis).Q  # [n, rows(R)], every column sums to 0
    t_small = (n / c) ** 0.5 * (basis @ r)  # [n, K + 1]
    t = torch.zeros((n, k_pad), dtype=torch.float64)
    t[:, :k_in] = t_small[:, :k_in]
    t[:, one_col] = t_small[:, k_in]
    return t.to(device=emb_w.device, dtype=dtype).contiguous()


def pack_bias(biases: Sequence[Optional[Tensor]], sizes: Sequence[int], device) -> Tensor:
    parts = [b.detach().float() if b is not None else torch.zeros(n, dtype=torch.float32, device=device)
             for b, n in zip(biases, sizes)]
    return torch.cat(parts, 0).contiguous() if len(parts) > 1 else parts[0].contiguous()


def f32c(t: Tensor) -> Tensor:
    """f32 contiguous view of a parameter (no copy for the usual f32 parameter)."""
    t = t.detach()
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


class GraphedForward:
    """One model forward captured in a HIP graph and replayed (``torch.cuda.CUDAGraph`` is ``hipGraph`` on ROCm).

    The forward is a fixed sequence of ~100 kernel launches through the C ABI; for the small configurations (O32 / O96
    grids) the step is bound by Python + launch latency, not by the GPU.  Capture needs nothing special from the
    kernels: they launch on torch's current stream, every buffer comes from torch's allocator (graph-private pool
    during capture), the edge plans / packed weights are cached by the warm-up calls, and no kernel of the path
    synchronises or allocates with hipMalloc.  Shapes and weights are frozen at capture time: re-capture after
    ``load_state_dict`` or for a different input shape.
    """

    def __init__(self, model, example_x: Tensor, warmup: int = 2) -> None:
        if not example_x.is_cuda:
            raise ValueError("GraphedForward: the example input must live on the GPU")
        self.model = model
        self.static_x = example_x.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():  # warm-up on a side stream (required before capture)
            for _ in range(max(1, warmup)):
                model(self.static_x)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.static_y = model(self.static_x)

    def __call__(self, x: Tensor) -> Tensor:
        if x.shape != self.static_x.shape or x.dtype != self.static_x.dtype:
            raise ValueError("GraphedForward: input shape / dtype differs from the captured one")
        self.static_x.copy_(x)
        self.graph.replay()
        return self.static_y


class DeviceDropout:
    """Attention-dropout seeds whose per-step part lives in DEVICE memory (reference ``layers/attention.py:90``: dropout of the
    attention probabilities in training mode, ``TransformerProcessor``'s default ``dropout_p = 0.1``).

    Eagerly every attention call draws its seed on the host (torch's CPU generator) and passes it as a kernel argument.  A
    training step captured in a HIP graph replays its kernel arguments, so the mask would repeat.  Inside this context the
    seed of a call is ``layer constant (host, drawn once per module) + word`` with ``word = step counter x odd constant``
    read by the kernels from device memory (``anemoi_mhsa``'s ``dropout_seed_dev``); :meth:`advance` -- two one-element
    kernels, capturable -- moves to the next step's masks.  ``GraphedTrainStep`` opens one for a model with dropout and
    makes ``advance`` the first captured operation; the same context around an eager loop (``advance()`` once per step)
    draws bit-identical masks, which is how the graphed step is tested against the eager one.

    A step's ``word`` is a tensor of its own (not updated in place), and the autograd node keeps the one its forward saw:
    a backward that runs after a later ``advance`` still rebuilds its own mask."""

    _active = None

    def __init__(self, device, start: int = 0) -> None:
        self.counter = torch.full((1,), int(start), dtype=torch.int64, device=device)
        self.word = self.counter * 0x9E3779B1

    def advance(self) -> None:
        self.counter.add_(1)
        self.word = self.counter * 0x9E3779B1  # (int64 arithmetic; the kernels read the low 32 bits)

    def pinned(self) -> "DeviceDropout":
        """A context with THIS step's ``word`` frozen: what a checkpointed region re-enters when it is recomputed -- inside or
        outside the original ``with`` block, before or after a later :meth:`advance` -- so that the recomputed forward draws
        the masks its first run drew (``training._checkpoint``)."""
        frozen = object.__new__(DeviceDropout)
        frozen.counter, frozen.word = self.counter, self.word
        return frozen

    def __enter__(self) -> "DeviceDropout":
        self._outer, DeviceDropout._active = DeviceDropout._active, self
        return self

    def __exit__(self, *exc) -> None:
        DeviceDropout._active = self._outer


def device_dropout() -> Optional[DeviceDropout]:
    """The :class:`DeviceDropout` context the caller runs in, or ``None`` (host-drawn seeds)."""
    return DeviceDropout._active


class GraphedTrainStep:
    """One TRAINING step -- forward, loss, backward, optionally the optimizer step -- captured in a HIP graph and replayed.

    The differentiable route issues several hundred launches per step from Python and from autograd's backward thread;
    below the N320 grid the step is bound by that host work, not by the GPU (config 2: 10 ms of forward enqueue for 10 ms
    of GPU work).  The capture follows torch's whole-network recipe: warm-up steps on a side stream, gradients set to
    ``None`` so that the captured backward allocates them from the graph's private pool, then every replay rewrites the
    same ``.grad`` tensors in place.  ``loss_fn(y, target)`` must be sync-free torch code; an ``optimizer`` must be
    capturable (``torch.optim.Adam(..., capturable=True)`` / SGD).  Without one the caller steps on ``param.grad`` after
    each call (and must not set the gradients to ``None``).  Shapes are frozen at capture time; parameters are read in
    place, so optimizer updates between replays are seen.

    torch's capture rule applies: no autograd graph over the model's parameters may be alive when this is built (drop
    references to the losses / outputs of earlier eager steps) -- a surviving graph keeps the parameters' gradient
    accumulators bound to the stream it ran on, the captured backward then touches that stream and the HIP runtime
    aborts the capture (observed as a crash in ``capture_end``).

    Attention dropout (``dropout_p > 0`` of the Transformer processor in training mode -- the reference's constructor
    default is 0.1, ``layers/processor.py:99``): the captured step runs inside a :class:`DeviceDropout` context
    (``self.dropout``) whose ``advance()`` is the first captured operation, so every replay draws new masks from the step
    counter in device memory; ``self.dropout.counter`` may be read or set between replays (reproducibility, resume).

    Side effects worth knowing: the ``warmup`` eager steps are REAL steps on ``example_x`` -- with an ``optimizer`` they
    update the parameters and the optimizer state (torch's capture recipe needs the optimizer's state initialised; pass
    ``warmup=1`` and a throw-away batch, or snapshot ``state_dict()`` around the constructor, if that matters).  The model's
    training / dropout configuration is frozen at capture: ``__call__`` refuses a model whose ``training`` flag or dropout
    rates have changed since.  The returned loss is a copy; ``static_y`` / ``static_loss`` alias graph memory the next
    replay overwrites.
    """

    def __init__(self, model, loss_fn, example_x: Tensor, example_target: Tensor, optimizer=None, warmup: int = 3) -> None:
        if not example_x.is_cuda:
            raise ValueError("GraphedTrainStep: the example input must live on the GPU")
        self.model, self.loss_fn, self.optimizer = model, loss_fn, optimizer
        self.static_x, self.static_target = example_x.clone(), example_target.clone()
        self._config = self._dropout_config()
        self.dropout = DeviceDropout(example_x.device) if model.training and any(p > 0.0 for p in self._config[1]) else None
        params = [p for p in model.parameters() if p.requires_grad]

        def zero():
            for p in params:
                p.grad = None

        def step(capturing: bool):
            if self.dropout is not None:
                self.dropout.advance()
            y = model(self.static_x)
            loss = loss_fn(y, self.static_target)
            loss.backward()
            if optimizer is not None:
                optimizer.step()
            if capturing:
                self.static_y, self.static_loss = y, loss

        import contextlib

        with self.dropout if self.dropout is not None else contextlib.nullcontext():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(max(1, warmup)):
                    zero()
                    step(False)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            zero()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                step(True)

    def _dropout_config(self):
        return bool(self.model.training), tuple(float(getattr(m, "dropout_p", 0.0) or 0.0) for m in self.model.modules())

    def __call__(self, x: Tensor, target: Tensor) -> Tensor:
        if x.shape != self.static_x.shape or target.shape != self.static_target.shape:
            raise ValueError("GraphedTrainStep: input / target shape differs from the captured one")
        if self._dropout_config() != self._config:
            raise RuntimeError("GraphedTrainStep: the model's training flag or dropout rates changed after the capture; "
                               "the graph still runs the captured configuration -- build a new GraphedTrainStep")
        self.static_x.copy_(x)
        self.static_target.copy_(target)
        self.graph.replay()
        return self.static_loss.detach().clone()
