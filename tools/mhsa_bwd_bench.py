#!/usr/bin/env python
"""Forward + backward of anemoi_mhsa alone (autograd.mhsa) at the Transformer-processor shapes of configs 2 and 3:
   python tools/mhsa_bwd_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anemoi_models_amd import autograd  # noqa: E402

shapes = [(10242, 16, 32), (40962, 16, 64)]
P_DROP = float(os.environ.get("MHSA_BENCH_DROPOUT", "0"))  # attention dropout (training mode of the reference; its default: 0.1)
if len(sys.argv) > 1:  # S,H,D triples
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for (s, h, d) in shapes:
    c = h * d
    x = (torch.randn(s, 3 * c, device="cuda") * 0.5).bfloat16().requires_grad_()
    dy = torch.randn(s, c, device="cuda").bfloat16()

    def fwd():
        with torch.no_grad():
            autograd.mhsa(x.detach(), 1, h, -1, P_DROP, 12345)

    def step():
        autograd.mhsa(x, 1, h, -1, P_DROP, 12345).backward(dy)
        x.grad = None

    res = []
    for fn in (fwd, step):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            fn()
        b.record()
        torch.cuda.synchronize()
        res.append(a.elapsed_time(b) / 3)
    flops = 4 * h * s * s * d
    print(f"dropout {P_DROP}: S={s} H={h} D={d}: forward {res[0]:.2f} ms, forward + backward {res[1]:.2f} ms -> backward {res[1] - res[0]:.2f} ms "
          f"({3.5 * flops / (res[1] - res[0]) / 1e9:.0f} TFLOP/s over its 7 S x S x D products)", flush=True)
