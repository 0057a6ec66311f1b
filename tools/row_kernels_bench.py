#!/usr/bin/env python
"""Stand-alone timings of the row kernels around the GEMMs at the config-3 data-grid size (542 080 nodes):
input assembly, the f32 decoder output GEMM (N = 80), output finalisation.   python tools/row_kernels_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anemoi_models_amd import ops, runtime  # noqa: E402

dev = torch.device("cuda", 0)


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


g, v = 542080, 90
x = torch.randn(1, 2, 1, g, v, device=dev)
ll = torch.randn(g, 4, device=dev)
tr = torch.randn(g, 8, device=dev)
for dtype, ld in ((torch.bfloat16, 256), (torch.float32, 224)):
    t = timed(lambda: ops.assemble_nodes(x, ll, tr, 1, dtype, ld_out=ld))
    nbytes = x.numel() * 4 + g * 12 * 4 + g * ld * torch.empty((), dtype=dtype).element_size()
    print(f"assemble_nodes [{g} x {ld}] {dtype}: {t * 1e3:.1f} us  ({nbytes / t / 1e9:.2f} TB/s)")
h = torch.randn(g, 1024, device=dev).bfloat16()
w = runtime.pack_weight([torch.randn(80, 1024, device=dev) / 32], torch.bfloat16)
bias = torch.randn(80, device=dev)
t = timed(lambda: ops.linear(h, w, bias, out_dtype=torch.float32))
print(f"linear [{g} x 1024] -> 80 (f32 out): {t * 1e3:.1f} us  ({(h.numel() * 2 + g * 80 * 4) / t / 1e9:.2f} TB/s)")
