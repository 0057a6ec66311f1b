#!/usr/bin/env python
"""Weight-gradient GEMM dW = dpre^T x at the config-3 shapes: the TN kernel (no transposed copies) against the older
route (chunked transposes + batched NT GEMM), same operands.   python tools/dw_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anemoi_models_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)


def timed(fn, iters=10):
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


shapes = [(40962, 4096, 1024), (40962, 1024, 4096), (40962, 4288, 1024), (40962, 1024, 1216), (542080, 1024, 256),
          (542080, 4096, 1024), (40962, 1024, 192), (5121, 4096, 1024)]
for m, n, k in shapes:
    torch.manual_seed(0)
    dpre = torch.randn(m, n, device=dev).bfloat16()
    x = torch.randn(m, k, device=dev).bfloat16()
    res = {}
    for name, tr in (("tn", False), ("transposes", True)):
        res[name] = (timed(lambda: ops.weight_grad(dpre, x, k, want_bias=True, transposed_route=tr)),
                     ops.weight_grad(dpre, x, k, transposed_route=tr))
    want = dpre[:8192].double().t() @ x[:8192].double() if m <= 8192 else None
    err = float((res["tn"][1] - res["transposes"][1]).abs().max() / res["transposes"][1].abs().max())
    fl = 2.0 * m * n * k / 1e9
    print(f"dW [{n} x {k}] over {m} rows: TN {res['tn'][0]:.3f} ms ({fl / res['tn'][0]:.0f} TFLOP/s incl. bias sums)   "
          f"transposes + NT {res['transposes'][0]:.3f} ms ({fl / res['transposes'][0]:.0f})   max |diff| / max = {err:.1e}")
    del dpre, x, res
