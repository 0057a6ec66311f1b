#!/usr/bin/env python
"""Coefficients of the packed-f32 GELU of csrc/gemm.hip::act_apply2: weighted minimax fit of (Phi(x) - 1/2) / x as a
polynomial in t = x^2 on |x| <= X (weight x^2: the quantity that matters is the GELU error |x| |Phi error|), converted
to the power basis in t and checked with an f32 Horner evaluation over [-9, 9] (clamped argument beyond X)."""
import sys

import numpy as np
from numpy.polynomial import chebyshev as C
from numpy.polynomial import polynomial as P
from scipy.special import erf

X = float(sys.argv[1]) if len(sys.argv) > 1 else 4.5
d = int(sys.argv[2]) if len(sys.argv) > 2 else 8


def phi(v):
    return 0.5 * (1 + erf(v / np.sqrt(2)))


x = np.cos(np.linspace(0, np.pi, 40001)) * X
x = x[np.abs(x) > 1e-9]
f, tt = (phi(x) - 0.5) / x, 2 * x * x / X**2 - 1
w = np.ones_like(x)
for _ in range(60):  # Lawson-style reweighting towards the minimax solution
    c = C.chebfit(tt, f, d, w=w * x * x)
    err = (C.chebval(tt, c) - f) * x * x
    w = w * (1 + 2 * np.abs(err) / np.abs(err).max())
pt = np.zeros(1)
for k, ck in enumerate(C.cheb2poly(c)):
    pt = P.polyadd(pt, ck * P.polypow([-1.0, 2 / X**2], k))
print("coefficients in t = x^2, highest first:")
for v in pt[::-1]:
    print(f"  {float(v)!r}f")
xs = np.linspace(-9, 9, 200001).astype(np.float32)
xc = np.clip(xs, -X, X).astype(np.float32)
t = (xc * xc).astype(np.float32)
q = np.full_like(xs, np.float32(pt[-1]))
for ck in pt[-2::-1]:
    q = (q * t + np.float32(ck)).astype(np.float32)
g = (xs * (xc * q + np.float32(0.5)).astype(np.float32)).astype(np.float32)
true = xs.astype(np.float64) * phi(xs.astype(np.float64))
e = np.abs(g - true)
print(f"X={X} degree {d}: max |GELU error| (f32 Horner) {e.max():.3e} at x={xs[e.argmax()]:.3f}; "
      f"max relative error for x > 0.05: {(e / np.maximum(np.abs(true), 1e-30))[xs > 0.05].max():.3e}")
