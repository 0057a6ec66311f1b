#!/usr/bin/env python
"""One training step (forward, loss, backward) of the flat GraphTransformer model on the HIP kernels through its nn.Module
at a bench workload:
   python tools/train_step_bench.py [cfg1|cfg2|cfg3] [steps] [GraphTransformer|GNN|Transformer]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
processor = sys.argv[3] if len(sys.argv) > 3 else "GraphTransformer"
os.environ.setdefault("ANEMOI_AMD_DTYPE", "bf16")
dev = torch.device("cuda", 0)
model, graph, x, _ = bench.build(workload, dev, processor)
model.train()
if os.environ.get("TRAIN_BENCH_DROPOUT"):  # attention dropout of the Transformer processor (reference default 0.1)
    for m in model.modules():
        if hasattr(m, "dropout_p"):
            m.dropout_p = float(os.environ["TRAIN_BENCH_DROPOUT"])
target = torch.zeros((1, 1, graph["data"].num_nodes, 80), device=dev)


def step():
    y = model(x)
    if os.environ.get("TRAIN_BENCH_PHASE") == "forward":  # profiling aid: the differentiable forward alone
        return 0.0
    loss = ((y - target) ** 2).mean()
    loss.backward()
    for p in model.parameters():
        p.grad = None
    return float(loss.detach())


if os.environ.get("TRAIN_BENCH_GRAPH") == "1":  # the whole step as one HIP graph (runtime.GraphedTrainStep)
    from anemoi_models_amd.runtime import GraphedTrainStep

    eager_loss = step()
    g = GraphedTrainStep(model, lambda y, t: ((y - t) ** 2).mean(), x, target)
    l0 = float(g(x, target))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        lg = g(x, target)
    host = (time.perf_counter() - t0) / steps * 1e3
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    layers = bench.WORKLOADS[workload][2]
    print(f"{workload} {processor} (HIP graph): forward + backward {ms:.1f} ms / step (host {host:.2f} ms) = "
          f"{graph['hidden'].num_nodes * layers / ms * 1e3:.3e} mesh-node updates/s (loss {float(lg):.6f}, eager {eager_loss:.6f})",
          flush=True)
    sys.exit(0)
step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss = step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
# where the step's wall time goes: host enqueue (the launches of the differentiable route are issued from Python) vs GPU
torch.cuda.synchronize()
t0 = time.perf_counter()
y = model(x)
t_host_f = time.perf_counter() - t0
torch.cuda.synchronize()
t_f = time.perf_counter() - t0
loss_t = ((y - target) ** 2).mean()
torch.cuda.synchronize()
t1 = time.perf_counter()
loss_t.backward()
t_host_b = time.perf_counter() - t1
torch.cuda.synchronize()
t_b = time.perf_counter() - t1
print(f"{workload} {processor}: differentiable forward {t_f * 1e3:.1f} ms (host enqueue {t_host_f * 1e3:.1f}), backward {t_b * 1e3:.1f} ms "
      f"(host enqueue {t_host_b * 1e3:.1f})", flush=True)
layers = bench.WORKLOADS[workload][2]
print(f"{workload} {processor}: forward + backward {ms:.1f} ms / step = {graph['hidden'].num_nodes * layers / ms * 1e3:.3e} mesh-node updates/s "
      f"(loss {loss:.4f}, peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB)", flush=True)
