"""Which rows differ between the four-wave kernel (fixed reference maximum) and the eight-wave kernel (exact online maximum, the
four-wave kernel's fallback) on the input of tools/micro/mhsa_repeat.py?  python tools/micro/mhsa_w4_vs_w8.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from anemoi_models_amd import ops
DEV = "cuda"
s, d, h = 40962, 64, 16
c = h * d
g = torch.Generator().manual_seed(s + d)
qkv = torch.randn(s, 3 * c, generator=g)
qkv[:, :c] *= 1.6
qkv = qkv.bfloat16().to(DEV)
a = ops.mhsa(qkv, 1, h, -1)       # four-wave kernel
b = ops.mhsa(qkv, 1, h, 10**6)    # eight-wave kernel (a window wider than the sequence masks nothing)
diff = (a.float() - b.float()).abs()
print(f"max |diff| {float(diff.max()):.3e} = {float(diff.max() / b.float().abs().max()):.2e} of the largest value")
for head in (0, 2, 1):
    rows = (a[:, head * d:(head + 1) * d] != b[:, head * d:(head + 1) * d]).any(1).nonzero().flatten()
    print(f"head {head}: {rows.numel()} rows differ; first {rows[:24].tolist()}; in rows 0..511: {int((rows < 512).sum())}")
