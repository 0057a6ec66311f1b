#!/usr/bin/env python
"""Lab: the MFMA tile edge kernel against the gather kernel (same inputs), correctness and time (GPU only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from anemoi_models_amd import ops, runtime  # noqa: E402
import binding, tiler  # noqa: E402
from anemoi_models_amd.graphs.synthetic import build_graph  # noqa: E402

dev = torch.device("cuda", 0)
gname = sys.argv[1] if len(sys.argv) > 1 else "n320_ico6"
c, h, up = (1024, 16, 12)
g = build_graph(gname)
lat, lon = g["hidden"].x[:, 0].double(), g["hidden"].x[:, 1].double()
inv = runtime.inverse_permutation(runtime.locality_order(torch.stack([lat.sin(), lon.sin(), lat.cos(), lon.cos()], 1)))
sets = {
    "proc": (torch.stack([inv[g[("hidden", "to", "hidden")].edge_index[0]], inv[g[("hidden", "to", "hidden")].edge_index[1]]]),
             g["hidden"].num_nodes, g["hidden"].num_nodes),
    "dec": (torch.stack([inv[g[("hidden", "to", "data")].edge_index[0]], g[("hidden", "to", "data")].edge_index[1]]),
            g["hidden"].num_nodes, g["data"].num_nodes),
    "enc": (torch.stack([g[("data", "to", "hidden")].edge_index[0], inv[g[("data", "to", "hidden")].edge_index[1]]]),
            g["data"].num_nodes, g["hidden"].num_nodes),
}
for name, (ei, n_src, n_dst) in sets.items():
    plan = runtime.build_edge_plan(ei.to(dev), n_src, n_dst)
    tiles = tiler.use_edge_mfma_tiles(plan, torch.bfloat16, c, h, up)
    use_tiles = tiles is not None and max(n_src, n_dst) * 2 * c * 2 < 2**31
    torch.manual_seed(0)
    kv = (torch.randn(n_src, 2 * c, device=dev) * 0.5).to(torch.bfloat16)
    sq = (torch.randn(n_dst, 2 * c + h * up, device=dev) * 0.5).to(torch.bfloat16)
    attr = torch.randn(plan.col.shape[0], up, device=dev)
    ld_out = ops.round_up(c + h * up, 64)
    args = (sq[:, c:2 * c], kv[:, :c], kv[:, c:], sq[:, :c], sq[:, 2 * c:], attr, plan.rowptr)
    lse0 = torch.empty(n_dst, h, device=dev)
    lse1 = torch.empty(n_dst, h, device=dev)
    want = ops.gt_edge_attention_folded(*args, plan.col, h, up, ld_out=ld_out, lse=lse0)
    print(f"{gname} {name}: gather checksum {float(want.float().sum()):.6e} {float(want.float().abs().sum()):.6e} lse {float(lse0[torch.isfinite(lse0)].sum()):.6e}")
    if not use_tiles:
        got = want.clone()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn = lambda: ops.gt_edge_attention_folded(*args, plan.col, h, up, out=want, ld_out=ld_out)
        for _ in range(3):
            fn()
        t0.record()
        for _ in range(20):
            fn()
        t1.record()
        torch.cuda.synchronize()
        alg = (2 * n_dst + 2 * n_src) * c * 2 + plan.col.shape[0] * 52 + (n_dst + 1) * 4
        ms = t0.elapsed_time(t1) / 20
        print(f"  gather {ms:.4f} ms  {alg / ms / 1e6:.0f} GB/s algorithmic ({alg / ms / 1e6 / 80:.1f} % of 8 TB/s)", flush=True)
        continue
    got = binding.gt_edge_attention_tiles(*args, tiles, h, up, ld_out=ld_out, lse=lse1)
    torch.cuda.synchronize()
    d = (got.float() - want.float()).abs()
    bad = ~torch.isfinite(got.float())
    print(f"{gname} {name}: caps ({tiles.src_cap}, {tiles.edge_cap}) tiles {tiles.n_tiles} reuse {tiles.reuse:.2f} fill {tiles.fill:.2f}; max |diff| out {float(d[:, :c].max()):.3e} "
          f"t {float(d[:, c:c + h * up].max()):.3e} (scale {float(want.float().abs().max()):.2f}), nan {int(bad.sum())}, "
          f"lse diff {float((lse0 - lse1).abs().max()):.3e}, rerun equal {bool(torch.equal(got, binding.gt_edge_attention_tiles(*args, tiles, h, up, ld_out=ld_out)))}",
          flush=True)
    if float(d.max()) > 0.1:
        rows = (d.max(1).values > 0.1).nonzero().flatten()
        print("  bad rows", rows[:16].tolist(), "of", rows.numel(), "cols", (d.max(0).values > 0.1).nonzero().flatten()[:16].tolist())
    alg = (2 * n_dst + 2 * n_src) * c * 2 + plan.col.shape[0] * 52 + (n_dst + 1) * 4
    for label, fn in (("gather", lambda: ops.gt_edge_attention_folded(*args, plan.col, h, up, out=want, ld_out=ld_out)),
                      ("tiles ", lambda: binding.gt_edge_attention_tiles(*args, tiles, h, up, out=got, ld_out=ld_out))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(20):
            fn()
        t1.record()
        torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / 20
        print(f"  {label} {ms:.4f} ms  {alg / ms / 1e6:.0f} GB/s algorithmic ({alg / ms / 1e6 / 80:.1f} % of 8 TB/s)", flush=True)
