"""Where the eager checkpointed training step of config 3 spends its wall time (allocator churn, host phases)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

os.environ.setdefault("ANEMOI_AMD_DTYPE", "bf16")
dev = torch.device("cuda", 0)
model, graph, x, _ = bench.build("cfg3", dev, "GraphTransformer")
model.train()
target = torch.zeros((1, 1, graph["data"].num_nodes, 80), device=dev)


def step(sync_mid=False):
    t0 = time.perf_counter()
    y = model(x)
    t1 = time.perf_counter()
    if sync_mid:
        torch.cuda.synchronize()
    loss = ((y - target) ** 2).mean()
    loss.backward()
    t2 = time.perf_counter()
    for p in model.parameters():
        p.grad = None
    v = float(loss.detach())
    t3 = time.perf_counter()
    return (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3


for _ in range(2):
    step()
for sync_mid in (False, True, False):
    s0 = torch.cuda.memory_stats()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rows = [step(sync_mid) for _ in range(4)]
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 4 * 1e3
    s1 = torch.cuda.memory_stats()
    print(f"sync_mid={sync_mid}: {ms:.1f} ms/step; host fwd/bwd/tail per step:",
          [tuple(round(a, 1) for a in r) for r in rows],
          "device allocs", s1["num_device_alloc"] - s0["num_device_alloc"], "frees", s1["num_device_free"] - s0["num_device_free"],
          "retries", s1["num_alloc_retries"] - s0["num_alloc_retries"], flush=True)
