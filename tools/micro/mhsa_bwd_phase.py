"""Shader-clock phases of one wave of the dK/dV kernel (library built with -DATT_BWD_PROF): python tools/micro/mhsa_bwd_phase.py LIB"""
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
names = ["loop top -> wait", "vmcnt wait", "barrier", "stage issue", "frag loads + S/dP MFMAs (+nops)", "softmax VALU + cvt", "acc MFMAs"]
v = list(out)[:7]
tot = sum(v)
for nm, t in zip(["(6 -> 0 wrap)"] + names[1:], v):
    print(f"{nm:36s} {t / n:9.1f} clocks / tile ({100 * t / tot:5.1f} %)")
print("total", tot / n, "s_memtime ticks per tile (100 MHz domain?)")
