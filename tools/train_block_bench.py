#!/usr/bin/env python
"""Forward + backward of ONE GraphTransformerProcessorBlock at the config-3 mesh size (40 962 nodes, 327 660 edges,
1024 channels, 16 heads) through anemoi_models_amd.autograd (random weights, synthetic graph); prints ms per training
step of the block and the forward-only time of the same differentiable path.   python tools/train_block_bench.py [bf16|fp32]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anemoi_models_amd import autograd, ops, runtime  # noqa: E402
from anemoi_models_amd.graphs.synthetic import build_graph  # noqa: E402

dtype = torch.float32 if (len(sys.argv) > 1 and sys.argv[1] == "fp32") else torch.bfloat16
dev = torch.device("cuda", 0)
g = build_graph("n320_ico6")
ei = g[("hidden", "to", "hidden")].edge_index
n, c, h, edge_dim = g["hidden"].num_nodes, 1024, 16, 11
lat, lon = g["hidden"].x[:, 0].double(), g["hidden"].x[:, 1].double()
order = runtime.locality_order(torch.stack([lat.sin(), lon.sin(), lat.cos(), lon.cos()], 1))
inv = runtime.inverse_permutation(order)
plan = runtime.build_edge_plan(torch.stack([inv[ei[0]], inv[ei[1]]]).to(dev), n, n)
up = ops.round_up(edge_dim + 1, 4)
torch.manual_seed(0)
attr = torch.randn(plan.col.shape[0], up, device=dev)
attr[:, edge_dim] = 1.0
attr[:, edge_dim + 1:] = 0.0


def lin(o, i):
    return {"weight": (torch.randn(o, i, device=dev) / i**0.5).requires_grad_(), "bias": torch.zeros(o, device=dev).requires_grad_()}


sd = {}
for name, (o, i) in {"lin_self": (c, c), "lin_query": (c, c), "lin_key": (c, c), "lin_value": (c, c), "lin_edge": (c, edge_dim),
                     "projection": (c, c), "node_dst_mlp.1": (4 * c, c), "node_dst_mlp.3": (c, 4 * c)}.items():
    for k, v in lin(o, i).items():
        sd[f"b.{name}.{k}"] = v
for name in ("layer_norm1", "node_dst_mlp.0"):
    sd[f"b.{name}.weight"] = torch.ones(c, device=dev).requires_grad_()
    sd[f"b.{name}.bias"] = torch.zeros(c, device=dev).requires_grad_()
x = torch.randn(n, c, device=dev).to(dtype).requires_grad_()
dz = torch.randn(n, c, device=dev).to(dtype)


def step(backward=True):
    z = autograd.gt_processor_block(x, sd, "b", attr, plan, h)
    if backward:
        z.backward(dz)
        x.grad = None
        for p in sd.values():
            p.grad = None


for label, bw in (("forward + backward", True), ("forward only (differentiable path)", False)):
    for _ in range(2):
        step(bw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        step(bw)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"{dtype}: {label}: {ms:.2f} ms per block = {n / ms * 1e3:.3e} mesh-node updates/s", flush=True)
