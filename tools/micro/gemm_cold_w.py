#!/usr/bin/env python
"""Does the staggered GEMM start gain on a COLD weight panel?  The same fc1-shaped GEMM (40 962 x 4096 x 1024,
GELU) back to back, (a) always on the same weight (L2 / MALL-warm after the first launch: tools/gemm_bench.py's situation, where
the stagger gains nothing) and (b) cycling through 48 different weights (384 MB: every launch finds its panel in HBM, as a
launch inside the model does), with ANEMOI_AMD_GEMM_STAGGER as set by the caller."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from anemoi_models_amd import ops  # noqa: E402

dev = "cuda"
m, n, k = 40962, 4096, 1024
x = torch.randn(m, k, device=dev).bfloat16()
ws = [(torch.randn(n, k, device=dev) / 32).bfloat16() for _ in range(48)]
b = torch.randn(n, device=dev)
out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)


def run(weights, its=96):
    for i in range(8):
        ops.linear(x, weights[i % len(weights)], b, act="GELU", out=out)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(its):
        ops.linear(x, weights[i % len(weights)], b, act="GELU", out=out)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / its * 1e3


for rep in range(3):
    print(f"stagger {os.environ.get('ANEMOI_AMD_GEMM_STAGGER', 'default')}: same weight {run(ws[:1]):7.1f} us per launch, "
          f"48 weights in turn {run(ws):7.1f} us", flush=True)
