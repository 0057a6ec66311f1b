#!/usr/bin/env python
"""A/B of the eight-wave attention kernel with the FULL wait states behind its S^T products (lab build
tools/micro/bin/libanemoi_amd_mhsa8pad.so, -DANEMOI_LAB_MHSA8_PAD=1) against the shipped form, in which hipcc leaves 5 ... 9
counted states on the paths across taken branches (tools/isa_hazard_audit.py --all: 20 pairs).  One process per library:
   python tools/micro/mhsa8_pad_ab.py <out.pt>            (through tools/micro/run_with_lib.py for the lab library)
saves the outputs of the shapes that take the eight-wave kernel (D = 64 with a window, D = 32 global and windowed) and prints
their timings; python tools/micro/mhsa8_pad_ab.py --compare a.pt b.pt  compares the bits."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

if sys.argv[1] == "--compare":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for key in a:
        same = torch.equal(a[key], b[key])
        print(f"{key}: {'bit-identical' if same else 'DIFFERENT: max abs ' + format(float((a[key].float() - b[key].float()).abs().max()), '.3e')}")
    sys.exit(0)

from anemoi_models_amd import ops  # noqa: E402

dev = "cuda"
res = {}
for s, h, d, window in ((40962, 16, 64, 1024), (40962, 16, 64, 4096), (10242, 16, 32, -1), (10242, 16, 32, 512), (40962, 16, 32, -1)):
    c = h * d
    qkv = torch.randn(s, 3 * c, generator=torch.Generator().manual_seed(s + d + window + 7)).bfloat16().to(dev)
    kw = {} if window < 0 else {"window": window}
    out = ops.mhsa(qkv, 1, h, **kw) if window < 0 else ops.mhsa(qkv, 1, h, window)
    for _ in range(3):
        ops.mhsa(qkv, 1, h) if window < 0 else ops.mhsa(qkv, 1, h, window)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    its = 20
    t0.record()
    for _ in range(its):
        y = ops.mhsa(qkv, 1, h) if window < 0 else ops.mhsa(qkv, 1, h, window)
    t1.record()
    torch.cuda.synchronize()
    same = torch.equal(y, out)
    res[f"S={s} H={h} D={d} window={window}"] = out.cpu()
    print(f"S={s} H={h} D={d} window={window}: {t0.elapsed_time(t1) / its:.4f} ms per call, repeat identical {same}", flush=True)
torch.save(res, sys.argv[1])
