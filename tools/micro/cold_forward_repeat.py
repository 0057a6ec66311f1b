#!/usr/bin/env python
"""Bit identity of a whole forward whose kernels start COLD (round 6: the one config-2 bf16 forward that moved inside a
whole-suite run came right behind ~10 s of CPU oracle work -- GPU idle, clocks down, instruction caches and L2 evicted --
while the 70 000 back-to-back repeats of tools/micro/forward_repeat.py always ran warm).

   python tools/micro/cold_forward_repeat.py cfg2 GraphTransformer 60 sleep|cpu|thrash|warm

Between forwards: ``sleep`` idles the GPU for a second, ``cpu`` runs f32 matmuls on every host thread for ~2 s (what the
oracle does to the box), ``thrash`` runs a few hundred unrelated torch kernels over 1 GiB (evicts the instruction caches
and the L2), ``f32`` runs an f32 forward of the same model object, ``warm`` does nothing (the control)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

os.environ.setdefault("ANEMOI_AMD_DTYPE", "bf16")
workload, processor = sys.argv[1], sys.argv[2]
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 60
modes = sys.argv[4].split(",") if len(sys.argv) > 4 else ["sleep", "cpu", "thrash", "f32", "warm"]
dev = torch.device("cuda", 0)
torch.set_num_threads(bench.host_threads())
model, graph, x, _ = bench.build(workload, dev, processor)
y0, l0 = bench.device_forward_with_latent(model, x)
y0, l0 = y0.clone(), l0.clone()
junk = torch.randn(256, 1024, 1024, device=dev)  # 1 GiB
host_a = torch.randn(2048, 2048)


def between(mode):
    if mode == "sleep":
        torch.cuda.synchronize()
        time.sleep(1.0)
    elif mode == "cpu":
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 2.0:
            host_a @ host_a
    elif mode == "f32":  # the suite alternates the two routes on one model object
        os.environ["ANEMOI_AMD_DTYPE"] = "fp32"
        with torch.no_grad():
            model(x)
        os.environ["ANEMOI_AMD_DTYPE"] = "bf16"
    elif mode == "thrash":
        for i in range(8):
            a = junk[i * 32:(i + 1) * 32]
            b = torch.sort(a.view(-1)[: 1 << 22]).values
            c = torch.nn.functional.gelu(a).sum() + b.cumsum(0)[-1] + torch.softmax(a[0], -1).amax()
            d = (a[0] @ a[1]).tanh().std() + c
            junk[0, 0, 0] = d * 0 + junk[0, 0, 0]
        torch.cuda.synchronize()


print(f"{workload} {processor}: checksum {float(y0.double().sum()):.6f} / {float(l0.double().sum()):.6f}", flush=True)
for mode in modes:
    bad_y = bad_l = 0
    notes = []
    t0 = time.perf_counter()
    for it in range(iters):
        between(mode)
        y, lat = bench.device_forward_with_latent(model, x)
        dl, dy = not torch.equal(lat, l0), not torch.equal(y, y0)
        bad_l += dl
        bad_y += dy
        if (dl or dy) and len(notes) < 6:
            ne_l = (lat != l0)
            ne_y = (y != y0).flatten(0, -2)
            notes.append(f"  repeat {it}: latent {int(ne_l.sum())} elements in {int(ne_l.any(1).sum())} rows "
                         f"(first row {int(ne_l.any(1).nonzero()[0]) if ne_l.any() else -1}, largest "
                         f"{float((lat.float() - l0.float()).abs().max() / l0.float().abs().max()):.2e}); prediction "
                         f"{int(ne_y.sum())} elements in {int(ne_y.any(1).sum())} rows, largest "
                         f"{float((y.float() - y0.float()).abs().max() / y0.float().abs().max()):.2e}")
    print(f"{mode:>6}: prediction differs in {bad_y} of {iters}, encoder latent in {bad_l}  ({time.perf_counter() - t0:.0f} s)",
          flush=True)
    for n in notes:
        print(n, flush=True)
