#!/usr/bin/env python
"""Micro-benchmark of the folded edge-attention kernel on the mesh / decoder / encoder graphs (GPU only)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anemoi_models_amd import ops, runtime  # noqa: E402
from anemoi_models_amd.graphs.synthetic import build_graph  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "proc"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
reorder = os.environ.get("REORDER", "1") == "1"
dev = "cuda"
g = build_graph(os.environ.get("GRAPH", "n320_ico6"))
key = {"proc": ("hidden", "to", "hidden"), "dec": ("hidden", "to", "data"), "enc": ("data", "to", "hidden")}[which]
ei = g[key].edge_index.to(dev)
n_src, n_dst = g[key[0]].num_nodes, g[key[2]].num_nodes
mesh_ll = g["hidden"].x
sc = torch.cat([torch.sin(mesh_ll), torch.cos(mesh_ll)], 1)
inv = runtime.inverse_permutation(runtime.locality_order(sc)).to(dev) if reorder else None
cache = runtime.PlanCache()
plan = cache.get(ei, n_src, n_dst, 1, None, inv if key[0] == "hidden" else None, inv if key[2] == "hidden" else None)
C, H, UP = 1024, 16, 12
dt = torch.bfloat16
q = torch.randn(n_dst, 2 * C + H * UP, device=dev).to(dt)   # x_r | q | u
kv = torch.randn(n_src, 2 * C, device=dev).to(dt)
ea = torch.randn(ei.shape[1], UP, device=dev)
out = torch.empty(n_dst, C + H * UP, device=dev, dtype=dt)


def run():
    ops.gt_edge_attention_folded(q[:, C:2 * C], kv[:, :C], kv[:, C:], q[:, :C], q[:, 2 * C:], ea, plan.rowptr, plan.col,
                                 H, UP, out=out)


for _ in range(3):
    run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(iters):
    run()
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / iters
alg = (2 * n_dst + 2 * n_src) * C * 2 + ei.shape[1] * 52 + (n_dst + 1) * 4
print(f"{which}: n_src={n_src} n_dst={n_dst} E={ei.shape[1]} reorder={reorder}  {ms:.4f} ms  {alg / ms / 1e6:.1f} GB/s algorithmic")
