#!/usr/bin/env python
"""Host-side enqueue time of one forward (no synchronisation inside the loop) vs its GPU time: tells when a rank becomes
launch-bound (at N GPUs the GPU time per rank shrinks ~N-fold, the enqueue time does not)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

os.environ.setdefault("ANEMOI_AMD_DTYPE", "bf16")
workload = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
dev = torch.device("cuda", 0)
model, graph, x, _ = bench.build(workload, dev)
with torch.no_grad():
    for _ in range(3):
        model(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        model(x)
    t_host = (time.perf_counter() - t0) / 10
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / 10
print(f"{workload}: host enqueue {t_host * 1e3:.2f} ms / forward, wall {t_all * 1e3:.2f} ms / forward")
if len(sys.argv) > 2:
    import cProfile
    import pstats

    pr = cProfile.Profile()
    with torch.no_grad():
        pr.enable()
        for _ in range(5):
            model(x)
        pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(25)
