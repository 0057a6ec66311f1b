"""Per-operand errors of the MFMA attention backward against torch autograd (f64): python tools/micro/mhsa_bwd_dbg.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from anemoi_models_amd import autograd  # noqa: E402

DEV = torch.device("cuda", 0)
for (b, s, h, d) in [(1, 64, 1, 64), (1, 128, 1, 64), (1, 96, 2, 64), (1, 700, 8, 64), (2, 333, 16, 32), (1, 4096, 2, 64)]:
    g = torch.Generator().manual_seed(s + d)
    c = h * d
    qkv = (torch.randn(b * s, 3 * c, generator=g) * 0.8).bfloat16()
    dout = torch.randn(b * s, c, generator=g).bfloat16()
    ref_in = qkv.double().requires_grad_()
    q, k, v = (t.reshape(b, s, h, d).permute(0, 2, 1, 3) for t in ref_in.split(c, dim=1))
    sc = q @ k.transpose(-1, -2) / d**0.5
    want = (torch.softmax(sc, -1) @ v).permute(0, 2, 1, 3).reshape(b * s, c)
    want.backward(dout.double())
    x = qkv.to(DEV).requires_grad_()
    got = autograd.mhsa(x, b, h, -1)
    got.backward(dout.to(DEV))
    gr = x.grad.double().cpu()
    errs = []
    for i, name in enumerate("qkv"):
        a, w = gr[:, i * c:(i + 1) * c], ref_in.grad[:, i * c:(i + 1) * c]
        e = (a - w).abs()
        rows = (e.max(1).values > 0.05 * w.abs().max()).nonzero().flatten()
        errs.append(f"d{name} {float(e.max() / w.abs().max()):.3e} bad rows {rows.numel()} {rows[:6].tolist()}..{rows[-3:].tolist()}")
    print((b, s, h, d), " | ".join(errs), "nan" if not torch.isfinite(gr).all() else "", flush=True)
    if (b, s, h, d) == (1, 64, 1, 64) and os.environ.get("DBG_DETAIL"):
        a, w = gr[:, :c], ref_in.grad[:, :c]
        e = (a - w).abs() / w.abs().max()
        print("dq err by (row block of 8) x (col block of 8):")
        print((e.reshape(8, 8, 8, 8).amax(dim=(1, 3)) * 100).round().int())
        print("got/want ratio sample row 0:", (a[0, :8] / w[0, :8]).tolist(), (a[0, 32:40] / w[0, 32:40]).tolist())
