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
    ap.add_argument("--set", default="proc", choices=["proc", "enc", "dec"],
                    help="edge set: mesh processor, encoder (grid -> mesh), decoder (mesh -> grid)")
    ap.add_argument("--save", default=None, help="write the output tensor here (bit-compare kernel variants)")
    ap.add_argument("--compare", default=None, help="compare the output with a tensor written by --save")
    a = ap.parse_args()
    if a.set != "proc":
        return mapper(a)
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

    sched = plan.schedule(torch.bfloat16, c) if os.environ.get("ANEMOI_AMD_EDGE_SCHED", "1") != "0" else None
    tiles = None
    if os.environ.get("ANEMOI_AMD_EDGE_TILES", "0") != "0":  # round 6: the LDS-tile kernel (caps: ANEMOI_AMD_EDGE_TILE_SRC / _EDGES)
        tiles = plan.tiles(torch.bfloat16, c, h, up)
        th = tiles.hdr.cpu()
        print(f"tiles: {tiles.n_tiles} (caps {tiles.src_cap} sources / {tiles.edge_cap} edges), {float(th[:, 5].float().mean()):.1f} "
              f"destinations, {float(th[:, 1].float().mean()):.0f} edges, {float(th[:, 3].float().mean()):.1f} distinct sources per "
              f"tile: a staged row serves {float(th[:, 1].sum()) / float(th[:, 3].sum()):.2f} edges")

    def run():
        ops.gt_edge_attention_folded(sq[:, c:2 * c], sq[:, 2 * c:3 * c], sq[:, 3 * c:4 * c], sq[:, :c], sq[:, 4 * c:],
                                     attr, plan.rowptr, plan.col, h, up, out=out, ld_out=ld_out, sched=sched, tiles=tiles)

    e = plan.col.shape[0]
    report(a, run, out, 4 * n * c * 2 + e * 52 + (n + 1) * 4, f"set=proc order={a.order} col={a.col} n={n} E={e}"
           + (f" tiles={tiles.src_cap}/{tiles.edge_cap}" if tiles is not None else ""))


def mapper(a):
    """Encoder / decoder edge sets at the model's own layouts: k|v = [n_src, 2C], x_r|q|u = [n_dst, 2C + H up]."""
    dev = torch.device("cuda", 0)
    g = build_graph(a.graph)
    key = ("data", "to", "hidden") if a.set == "enc" else ("hidden", "to", "data")
    ei = g[key].edge_index
    n_src, n_dst = g[key[0]].num_nodes, g[key[2]].num_nodes
    lat, lon = g["hidden"].x[:, 0].double(), g["hidden"].x[:, 1].double()
    inv = runtime.inverse_permutation(runtime.locality_order(torch.stack([lat.sin(), lon.sin(), lat.cos(), lon.cos()], 1)))
    src, dst = (ei[0], inv[ei[1]]) if a.set == "enc" else (inv[ei[0]], ei[1])
    plan = runtime.build_edge_plan(torch.stack([src, dst]).to(dev), n_src, n_dst)
    c, h, up = a.channels, a.heads, 12
    torch.manual_seed(0)
    kv = (torch.randn(n_src, 2 * c, device=dev) * 0.5).to(torch.bfloat16)
    sq = (torch.randn(n_dst, 2 * c + h * up, device=dev) * 0.5).to(torch.bfloat16)
    attr = torch.randn(plan.col.shape[0], up, device=dev)
    ld_out = ops.round_up(c + h * up, 64)
    out = torch.zeros(n_dst, ld_out, dtype=torch.bfloat16, device=dev)

    runs = plan.runs3() if os.environ.get("ANEMOI_AMD_EDGE_RUNS", "1") != "0" else None  # decoder: shared-source runs
    if runs is not None:
        print(f"runs: {runs[0].shape[0] - 1} for {n_dst} destinations (mean length {n_dst / (runs[0].shape[0] - 1):.2f})")

    sched = plan.schedule(torch.bfloat16, c) if os.environ.get("ANEMOI_AMD_EDGE_SCHED", "1") != "0" and runs is None else None

    def run():
        ops.gt_edge_attention_folded(sq[:, c:2 * c], kv[:, :c], kv[:, c:], sq[:, :c], sq[:, 2 * c:], attr, plan.rowptr,
                                     plan.col, h, up, out=out, ld_out=ld_out, runs=runs, sched=sched)

    report(a, run, out, (2 * n_dst + 2 * n_src) * c * 2 + plan.col.shape[0] * 52 + (n_dst + 1) * 4,
           f"set={a.set} n_src={n_src} n_dst={n_dst} E={plan.col.shape[0]}")


def report(a, run, out, alg, label):
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
    note = ""
    if a.save:
        torch.save(out.cpu(), a.save)
    if a.compare:
        want = torch.load(a.compare)
        note = "  bit-identical to " + a.compare if torch.equal(out.cpu(), want) else \
            f"  DIFFERS from {a.compare}: max abs {float((out.cpu().float() - want.float()).abs().max()):.3e}"
    label += f" sched={os.environ.get('ANEMOI_AMD_EDGE_SCHED', '1')} U={os.environ.get('ANEMOI_AMD_EDGE_U', '4')}"
    print(f"{label} {ms:.4f} ms  {alg / ms / 1e6:.0f} GB/s "
          f"algorithmic ({alg / ms / 1e6 / 80:.1f} % of 8 TB/s){note}", flush=True)


if __name__ == "__main__":
    main()
