#!/usr/bin/env python
"""Lab: per-phase shader clocks of one wave of the MFMA tile edge kernel (ET_PROF build only)."""
import ctypes, os, sys
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
ei = torch.stack([inv[g[("hidden", "to", "hidden")].edge_index[0]], inv[g[("hidden", "to", "hidden")].edge_index[1]]])
n = g["hidden"].num_nodes
plan = runtime.build_edge_plan(ei.to(dev), n, n)
tiles = tiler.use_edge_mfma_tiles(plan, torch.bfloat16, c, h, up)
kv = (torch.randn(n, 2 * c, device=dev) * 0.5).to(torch.bfloat16)
sq = (torch.randn(n, 2 * c + h * up, device=dev) * 0.5).to(torch.bfloat16)
attr = torch.randn(plan.col.shape[0], up, device=dev)
ld_out = ops.round_up(c + h * up, 64)
args = (sq[:, c:2 * c], kv[:, :c], kv[:, c:], sq[:, :c], sq[:, 2 * c:], attr, plan.rowptr)
lib = binding.lib()
for it in range(3):
    binding.gt_edge_attention_tiles(*args, tiles, h, up, ld_out=ld_out)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    lib.lab_edge_prof(buf)
    v = list(buf)
    iters = max(v[8], 1)
    names = ["prefetch issue", "C build", "S mfma", "softmax+alpha", "alpha pack", "vmcnt wait", "PV+stage+store", "t loop"]
    print(f"run {it}: iterations {iters}, total/iter {sum(v[:8]) / iters:.0f} clocks")
    for nm, x in zip(names, v[:8]):
        print(f"   {nm:16s} {x / iters:8.0f}")
