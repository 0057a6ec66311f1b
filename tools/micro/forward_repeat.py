#!/usr/bin/env python
"""Run-to-run bit identity of a WHOLE forward (prediction and encoder latent) under whatever else shares the GPU:
   python tools/micro/forward_repeat.py cfg2 GraphTransformer 300
Start two of them at once for contention (round 5 found its attention hazard that way; round 6 looks for whatever moved the
config-2 bf16 latent once inside a whole-suite run)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

os.environ.setdefault("ANEMOI_AMD_DTYPE", "bf16")
workload, processor = sys.argv[1], sys.argv[2]
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 300
dev = torch.device("cuda", 0)
model, graph, x, _ = bench.build(workload, dev, processor)
y0, l0 = bench.device_forward_with_latent(model, x)
y0, l0 = y0.clone(), l0.clone()
print("started", flush=True)
bad_y = bad_l = 0
worst = 0.0
for it in range(iters):
    y, lat = bench.device_forward_with_latent(model, x)
    if not torch.equal(lat, l0):
        bad_l += 1
    if not torch.equal(y, y0):
        bad_y += 1
        worst = max(worst, float((y.float() - y0.float()).abs().max() / y0.float().abs().max()))
print(f"{workload} {processor}: prediction differs in {bad_y} of {iters} repeats (largest relative difference {worst:.2e}), "
      f"encoder latent in {bad_l}; checksum {float(y0.double().sum()):.6f} / {float(l0.double().sum()):.6f}", flush=True)
