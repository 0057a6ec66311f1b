#!/usr/bin/env python
"""What can a dataflow-chained persistent GEMM recover at the per-rank shapes?  (round 6, VERDICT r05 item 1 -- lab tool)

A chain of Linears in ONE persistent launch removes, per seam, (a) the drain of the producer launch (its last tiles'
epilogues with nothing staged behind them), (b) the launch boundary and (c) the fill of the consumer launch (tail-row pass,
first slab's latency) -- and ADDS the dependency wait of every consumer tile on its producer row block.  (a) + (b) + (c) is
measurable WITHOUT writing the chain: the persistent kernel already walks the tiles of several independent problems as one
list (``anemoi_linear_batched``), so

    seam = t(two launches of one problem each) - t(one launch of both problems)

at a tile count that fills the chip in whole rounds both ways (no quantisation difference) is the whole fixed cost of a seam,
i.e. an UPPER BOUND on what a chain of dependent problems can save per seam (its consumers additionally wait for data).
Printed next to the step times of the per-rank pair fc1 -> fc2 (M = 5 121) it bounds."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from anemoi_models_amd import _lib, ops  # noqa: E402

dev = "cuda"
lib = _lib.load()
BF = ops.dtype_code(torch.bfloat16)


def batched(x, w, y, batch):
    b, m, k = x.shape
    n = w.shape[1]
    st = lib.anemoi_linear_batched(BF, BF, x.data_ptr(), k, m * k, w.data_ptr(), n * k, y.data_ptr(), n, m * n, batch, m, n, k,
                                   ops._stream())
    _lib.check(st, "anemoi_linear_batched")


def timeit(fn, it=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3  # us


print("seam = 2 x t(launch of B problems) - t(launch of 2B problems); tiles per launch a multiple of 256 both ways")
for m, n, k, b in [(4096, 4096, 1024, 1), (4096, 1024, 4096, 4), (4096, 2048, 1024, 2), (4096, 1024, 1024, 4),
                   (2048, 4096, 1024, 2)]:
    x = torch.randn(2 * b, m, k, device=dev).bfloat16()
    w = (torch.randn(2 * b, n, k, device=dev) / k**0.5).bfloat16()
    y = torch.empty(2 * b, m, n, device=dev, dtype=torch.bfloat16)
    tiles = (m // 256) * (n // 256) * b
    t1 = timeit(lambda: (batched(x[:b], w[:b], y[:b], b), batched(x[b:], w[b:], y[b:], b)))
    t2 = timeit(lambda: batched(x, w, y, 2 * b))
    print(f"M={m} N={n} K={k}  {b} problem(s) = {tiles} tiles per launch: two launches {t1:7.1f} us, one launch of both "
          f"{t2:7.1f} us, seam {t1 - t2:5.1f} us = {100 * (t1 - t2) / t1:4.1f} % of the pair", flush=True)

# the pair the stop-loss is stated on (fc1 -> GELU -> fc2 at a rank's 5 121 rows), as the model launches it
m = 5121
x = torch.randn(m, 1024, device=dev).bfloat16()
w1 = (torch.randn(4096, 1024, device=dev) / 32).bfloat16()
w2 = (torch.randn(1024, 4096, device=dev) / 64).bfloat16()
b1, b2 = torch.randn(4096, device=dev), torch.randn(1024, device=dev)
h = torch.empty(m, 4096, device=dev, dtype=torch.bfloat16)
y = torch.empty(m, 1024, device=dev, dtype=torch.bfloat16)
t_a = timeit(lambda: ops.linear(x, w1, b1, act="GELU", out=h))
t_b = timeit(lambda: ops.linear(h, w2, b2, residual=x, out=y))
t_ab = timeit(lambda: (ops.linear(x, w1, b1, act="GELU", out=h), ops.linear(h, w2, b2, residual=x, out=y)))
print(f"M=5121 fc1 (4096 x 1024, GELU) {t_a:6.1f} us, fc2 (1024 x 4096, + residual) {t_b:6.1f} us, the pair back to back "
      f"{t_ab:6.1f} us; the stop-loss asks for <= {0.92 * t_ab:6.1f} us (-8 %)")
