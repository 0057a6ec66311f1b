#!/usr/bin/env python
"""Mesh edge-phase kernel alone at the config-3 processor shapes (for rocprofv3 PMC passes and knob A/B):
   python tools/edge_bench.py [--iters 20] [--order morton|natural] [--graph n320_ico6] [--channels 1024]
Prints avg ms per launch and algorithmic GB/s.  Knobs are the kernel's env variables (ANEMOI_AMD_EDGE_*)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anemoi_models_amd import ops, runtime  # noqa: E402
from anemoi_models_amd.graphs.synthetic import build_graph  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--order", default="morton")
    ap.add_argument("--graph", default="n320_ico6")
    ap.add_argument("--channels", type=int, default=1024)
    ap.add_argument("--heads", type=int, default=16)
    ap.add_argument("--col", default="graph", help="graph | near (sources = dst-4..dst+4: ideal locality) | random")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    g = build_graph(a.graph)
    ei = g[("hidden", "to", "hidden")].edge_index
    n = g["hidden"].num_nodes
    lat, lon = g["hidden"].x[:, 0].double(), g["hidden"].x[:, 1].double()
    sincos = torch.stack([lat.sin(), lon.sin(), lat.cos(), lon.cos()], 1)
    if a.order == "natural":
        order = torch.arange(n)
    elif a.order == "morton":
        order = runtime.locality_order(sincos)
    else:
        order = getattr(runtime, "locality_order_" + a.order)(sincos)
    inv = runtime.inverse_permutation(order)
    plan = runtime.build_edge_plan(torch.stack([inv[ei[0]], inv[ei[1]]]).to(dev), n, n)
    if a.col != "graph":
        dst = torch.repeat_interleave(torch.arange(n, device=dev), (plan.rowptr[1:] - plan.rowptr[:-1]).long())
        k_in_row = torch.arange(plan.col.shape[0], device=dev) - plan.rowptr[:-1].long()[dst]
        if a.col == "near":
            plan.col = (dst + k_in_row - 4).clamp_(0, n - 1).to(torch.int32)
        else:
            plan.col = torch.randint(0, n, plan.col.shape, device=dev, dtype=torch.int32)
    c, h, up = a.channels, a.heads, 12
    torch.manual_seed(0)
    sq = (torch.randn(n, 4 * c + h * up, device=dev) * 0.5).to(torch.bfloat16)
    attr = torch.randn(plan.col.shape[0], up, device=dev)
    ld_out = ops.round_up(c + h * up, 64)
    out = torch.zeros(n, ld_out, dtype=torch.bfloat16, device=dev)

    def run():
        ops.gt_edge_attention_folded(sq[:, c:2 * c], sq[:, 2 * c:3 * c], sq[:, 3 * c:4 * c], sq[:, :c], sq[:, 4 * c:],
                                     attr, plan.rowptr, plan.col, h, up, out=out, ld_out=ld_out)

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(a.iters):
        run()
    t1.record()
    torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / a.iters
    e = plan.col.shape[0]
    alg = 4 * n * c * 2 + e * 52 + (n + 1) * 4
    print(f"order={a.order} col={a.col} n={n} E={e} {ms:.4f} ms  {alg / ms / 1e6:.0f} GB/s algorithmic "
          f"({alg / ms / 1e6 / 80:.1f} % of 8 TB/s)", flush=True)


if __name__ == "__main__":
    main()
