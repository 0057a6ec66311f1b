#!/usr/bin/env python
"""Repeat the bf16 MFMA attention forward on a few ragged shapes with fresh inputs and shifting allocations, against a
torch f32 reference: a hunt for timing / address dependent errors.   python tools/mhsa_stress.py [iterations]"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anemoi_models_amd import ops  # noqa: E402

dev = "cuda"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
random.seed(0)
for (b, s, h, d, w) in [(2, 1111, 16, 32, -1), (2, 333, 16, 32, -1), (2, 1000, 16, 64, -1), (1, 900, 4, 32, 70), (3, 200, 16, 32, -1)]:
    c = h * d
    bad, worst = 0, 0.0
    keep = []
    for it in range(iters):
        junk = [torch.full((random.randint(1, 1 << 20),), float("nan"), device=dev) for _ in range(random.randint(0, 3))]
        if it % 7 == 0:
            keep = junk  # hold some allocations so that later tensors land elsewhere
        g = torch.Generator().manual_seed(it)
        qkv = (torch.randn(b * s, 3 * c, generator=g) * 0.8).bfloat16().to(dev)
        del junk
        out = ops.mhsa(qkv, b, h, w)
        q, k, v = (t.float().reshape(b, s, h, d).permute(0, 2, 1, 3) for t in qkv.split(c, dim=1))
        sc = q @ k.transpose(-1, -2) / d**0.5
        if w >= 0:
            i = torch.arange(s, device=dev)
            sc = sc.masked_fill((i[:, None] - i[None, :]).abs() > w, float("-inf"))
        want = (torch.softmax(sc, -1) @ v).permute(0, 2, 1, 3).reshape(b * s, c)
        err = float((out.float() - want).abs().max() / want.abs().max())
        if not err < 2e-2:
            bad += 1
            diff = (out.float() - want).abs().reshape(b, s, h, d)
            bb, ss, hh, _ = [int(x[0]) for x in torch.nonzero(diff == diff.max())[:1].t()] if err == err else (-1, -1, -1, -1)
            print(f"  iteration {it}: rel err {err:.3e} at batch {bb}, query {ss}, head {hh}; nan rows "
                  f"{int(torch.isnan(out.float()).any(dim=1).sum())}", flush=True)
        worst = max(worst, err if err == err else 9.9)
    print(f"B={b} S={s} H={h} D={d} window={w}: {bad} bad of {iters}, worst {worst:.3e}", flush=True)
