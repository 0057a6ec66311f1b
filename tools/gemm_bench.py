#!/usr/bin/env python
"""Micro-benchmark of anemoi_linear on the shapes of the forward path (GPU only; not part of the product)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--lib" in sys.argv:  # lab builds of the kernel library (tools/micro/bin/*.so), e.g. the tail-epilogue GEMM for A/B runs
    _i = sys.argv.index("--lib")
    from anemoi_models_amd import _lib as _lab_lib

    _lab_lib.LIB_PATH = os.path.abspath(sys.argv[_i + 1])  # ANEMOI_LAB_LIB
    del sys.argv[_i:_i + 2]
from anemoi_models_amd import ops  # noqa: E402

SHAPES = [(40962, 4096, 1024), (40962, 1024, 4096), (40962, 1024, 1024), (542080, 4096, 1024),
          (542080, 1024, 4096), (542080, 2048, 1024), (542080, 1024, 192), (40960, 4096, 256), (40960, 4096, 512),
          (40960, 4096, 2048), (40960, 4096, 8192)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
dev = "cuda"
for m, n, k in SHAPES:
    x = torch.randn(m, k, device=dev).bfloat16()
    w = (torch.randn(n, k, device=dev) / k**0.5).bfloat16()
    b = torch.randn(n, device=dev)
    r = torch.randn(m, n, device=dev).bfloat16()
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    if os.environ.get("GEMM_BENCH_BLASLT", "1") != "0":  # library yardstick: torch.nn.functional.linear -> hipBLASLt
        bb = b.bfloat16()
        for _ in range(3):
            torch.nn.functional.linear(x, w, bb)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            torch.nn.functional.linear(x, w, bb)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 10
        print(f"M={m:7d} N={n:5d} K={k:5d} hipBLASLt (torch F.linear + bias)  {ms:8.4f} ms  "
              f"{2 * m * n * k / ms / 1e9:8.1f} TFLOP/s", flush=True)
    cases = [("Identity", None, None), ("GELU", None, None), ("Identity", r, None)]
    if os.environ.get("GEMM_BENCH_STATS", "0") != "0":  # + the row-statistics epilogue (anemoi_linear_stats: split-K route)
        cases += [("Identity", None, 1e-5), ("Identity", r, 1e-5)]
    for act, res, eps in cases:
        for _ in range(3):
            ops.linear(x, w, b, act=act, residual=res, out=out, stats_eps=eps)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        it = 10
        for _ in range(it):
            ops.linear(x, w, b, act=act, residual=res, out=out, stats_eps=eps)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / it
        print(f"M={m:7d} N={n:5d} K={k:5d} act={act:8s} res={res is not None!s:5s} stats={eps is not None!s:5s} {ms:8.4f} ms  "
              f"{2 * m * n * k / ms / 1e9:8.1f} TFLOP/s", flush=True)
