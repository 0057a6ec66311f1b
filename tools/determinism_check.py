#!/usr/bin/env python
"""Run-to-run determinism of the forward (and of one training step) under repetition with shifting allocations: every
kernel of the path is atomics-free, so any difference between two runs of the same input is a race.
python tools/determinism_check.py [cfg2|cfg3] [repeats]"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 40
os.environ.setdefault("ANEMOI_AMD_DTYPE", "bf16")
dev = torch.device("cuda", 0)
random.seed(0)
for processor in ("GraphTransformer", "GNN", "Transformer"):
    model, graph, x, _ = bench.build(workload, dev, processor)
    model.eval()
    first, bad = None, 0
    for it in range(repeats):
        junk = [torch.full((random.randint(1, 1 << 22),), float("nan"), device=dev) for _ in range(random.randint(0, 3))]
        with torch.no_grad():
            y = model(x)
        del junk
        if first is None:
            first = y.clone()
        elif not torch.equal(first, y):
            bad += 1
            d = (first - y).abs()
            print(f"  {processor} run {it}: max |diff| {float(d.max()):.3e} in {int((d > 0).sum())} values", flush=True)
    print(f"{workload} {processor}: {bad} of {repeats - 1} repeated forwards differ from the first; finite: "
          f"{bool(torch.isfinite(first).all())}", flush=True)
    if processor == "GraphTransformer":
        model.train()
        grads = None
        bad = 0
        for it in range(6):
            junk = [torch.full((random.randint(1, 1 << 22),), float("nan"), device=dev) for _ in range(random.randint(0, 2))]
            model(x).float().pow(2).mean().backward()
            del junk
            g = torch.cat([p.grad.flatten() for p in model.parameters() if p.grad is not None])
            for p in model.parameters():
                p.grad = None
            if grads is None:
                grads = g.clone()
            elif not torch.equal(grads, g):
                bad += 1
        print(f"{workload} {processor} training step: {bad} of 5 repeated gradient sets differ from the first", flush=True)
    del model
    torch.cuda.empty_cache()
