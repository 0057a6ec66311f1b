#!/usr/bin/env python
"""anemoi_mhsa alone at the Transformer-processor shape of config 3 (S = 40 962 mesh nodes, 16 heads, D = 64, bf16):
   python tools/mhsa_bench.py [S] [H] ; also checks the result of a 2048-query slice against an f32 torch reference."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anemoi_models_amd import ops  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 40962
H = int(sys.argv[2]) if len(sys.argv) > 2 else 16
D = int(sys.argv[3]) if len(sys.argv) > 3 else 64
dev = "cuda"
C = H * D
torch.manual_seed(0)
qkv = (torch.randn(S, 3 * C, device=dev) * 1.0).bfloat16()
out = ops.mhsa(qkv, 1, H)
torch.cuda.synchronize()
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(5):
    ops.mhsa(qkv, 1, H)
t1.record()
torch.cuda.synchronize()
ms = t0.elapsed_time(t1) / 5
print(f"S={S} H={H} D={D}: {ms:.3f} ms  {4 * H * S * S * D / ms / 1e9:.1f} TFLOP/s", flush=True)
# reference on a slice of queries (head 0 and the last head), f32
for h in (0, H - 1):
    q = qkv[:2048, h * D:(h + 1) * D].float()
    k = qkv[:, C + h * D:C + (h + 1) * D].float()
    v = qkv[:, 2 * C + h * D:2 * C + (h + 1) * D].float()
    want = torch.softmax(q @ k.T / D**0.5, dim=-1) @ v
    got = out[:2048, h * D:(h + 1) * D].float()
    print(f"head {h}: max abs err {float((got - want).abs().max()):.4e} (|want| max {float(want.abs().max()):.3f})")
q = qkv[S - 100:, :D].float()
want = torch.softmax(q @ qkv[:, C:C + D].float().T / D**0.5, dim=-1) @ qkv[:, 2 * C:2 * C + D].float()
print(f"last queries: max abs err {float((out[S - 100:, :D].float() - want).abs().max()):.4e}")
