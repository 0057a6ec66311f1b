"""Shader-clock phases of one wave of the dK/dV kernel (library built with -DATT_BWD_PROF; the marks cost an lgkmcnt(0) each, so LDS latency shows up in the phase that ends at the next mark):
   python tools/micro/mhsa_bwd_phase.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from anemoi_models_amd import autograd, _lib  # noqa: E402

s, h, d = 40962, 16, 64
c = h * d
x = (torch.randn(s, 3 * c, device="cuda") * 0.5).bfloat16().requires_grad_()
dy = torch.randn(s, c, device="cuda").bfloat16()
for _ in range(2):
    autograd.mhsa(x, 1, h, -1).backward(dy)
    x.grad = None
torch.cuda.synchronize()
lib = _lib.load()
out = (ctypes.c_ulonglong * 16)()
lib.anemoi_debug_att_prof.argtypes = [ctypes.c_void_p]
print("rc", lib.anemoi_debug_att_prof(out))
n = (s + 31) // 32
names = ["(6 -> 0 wrap)", "vmcnt wait", "barrier", "stage issue", "M1 (frag loads, S/dP MFMAs)", "V (softmax, cvt)", "M2 (acc MFMAs)", "-"]
v = list(out)[:8]
tot = sum(v)
for nm, t in zip(names, v):
    print(f"{nm:36s} {t / n:9.1f} clocks / tile ({100 * t / tot:5.1f} %)")
print("total", round(tot / n, 1), "clocks per tile")
