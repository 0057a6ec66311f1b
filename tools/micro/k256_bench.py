"""K = 256 products of the mappers at the config-3 grid size: python tools/micro/k256_bench.py  (ANEMOI_AMD_GEMM_KSTREAM=0|1)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from anemoi_models_amd import ops
dev = "cuda"
m, k = 542080, 256
x = torch.randn(m, k, device=dev).bfloat16()
for n, fold in ((2048, True), (2240, True), (1024, False), (256, False)):
    w = (torch.randn(n, k, device=dev) / 16).bfloat16()
    b = torch.randn(n, device=dev)
    ln = (torch.rand(m, 2, device=dev).contiguous(), torch.randn(n, device=dev)) if fold else None
    y = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    for _ in range(3): ops.linear(x, w, b, ln=ln, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.linear(x, w, b, ln=ln, out=y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"K=256 M={m} N={n} fold={fold}: {ms:.4f} ms  {2*m*n*k/ms/1e9:.0f} TFLOP/s  output {m*n*2/ms/1e6:.0f} GB/s", flush=True)
