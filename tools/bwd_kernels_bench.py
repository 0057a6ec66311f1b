#!/usr/bin/env python
"""Stand-alone timings of the backward-side kernels at the config-3 mesh size (40 962 nodes, 327 660 edges, 1024 ch):
the transposes / column sums of the weight-gradient path against a plain device copy, and the three edge-backward
kernels.   python tools/bwd_kernels_bench.py [transpose|edge|all]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anemoi_models_amd import autograd, ops, runtime  # noqa: E402

dev = torch.device("cuda", 0)
what = sys.argv[1] if len(sys.argv) > 1 else "all"


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


if what in ("transpose", "all"):
    for rows, cols in ((40962, 1024), (40962, 4096), (542080, 1024), (40962, 192)):
        x = torch.randn(rows, cols, device=dev).bfloat16()
        y = torch.empty_like(x)
        gb = 2 * x.numel() * 2 / 1e6  # GB moved per call / time in ms = TB/s
        t_copy = timed(lambda: y.copy_(x))
        t_tr = timed(lambda: ops.transpose(x, ops.round_up(rows, 64)))
        t_torch = timed(lambda: x.t().contiguous())
        t_cs = timed(lambda: ops.col_sum(x))
        print(f"[{rows} x {cols}] bf16: copy {t_copy * 1e3:7.1f} us ({gb / t_copy / 1e3:4.1f} TB/s)  ops.transpose {t_tr * 1e3:7.1f} us "
              f"({gb / t_tr / 1e3:4.1f} TB/s)  torch t().contiguous() {t_torch * 1e3:7.1f} us  col_sum {t_cs * 1e3:7.1f} us "
              f"({gb / 2 / t_cs / 1e3:4.1f} TB/s)")

if what in ("edge", "all"):
    from anemoi_models_amd.graphs.synthetic import build_graph

    g = build_graph("n320_ico6")
    ei = g[("hidden", "to", "hidden")].edge_index
    n, c, h, up = g["hidden"].num_nodes, 1024, 16, 12
    lat, lon = g["hidden"].x[:, 0].double(), g["hidden"].x[:, 1].double()
    order = runtime.locality_order(torch.stack([lat.sin(), lon.sin(), lat.cos(), lon.cos()], 1))
    inv = runtime.inverse_permutation(order)
    plan = runtime.build_edge_plan(torch.stack([inv[ei[0]], inv[ei[1]]]).to(dev), n, n)
    torch.manual_seed(0)
    attr = torch.randn(plan.col.shape[0], up, device=dev)
    sq = torch.randn(n, 4 * c + h * up, device=dev).bfloat16().requires_grad_()
    dfull = torch.randn(n, c + h * up, device=dev).bfloat16()

    def fwd_bwd():
        out = autograd._GTEdgeAttentionSelf.apply(sq, attr, plan, h, up)
        out.backward(dfull)
        sq.grad = None

    def fwd():
        with torch.no_grad():
            autograd._GTEdgeAttentionSelf.apply(sq, attr, plan, h, up)

    t_f, t_fb = timed(fwd), timed(fwd_bwd)
    print(f"edge phase, mesh block ({n} nodes, {plan.col.shape[0]} edges): forward {t_f * 1e3:.1f} us, forward + backward "
          f"{t_fb * 1e3:.1f} us -> backward {1e3 * (t_fb - t_f):.1f} us (destination-major + source-major + glue)")
