#!/usr/bin/env python
"""What would the ONE-PASS attention backward of the review's item 6 pay for its dQ partials?  (round 6, measured instead of
built -- as tools/micro/chain_upper_bound.py did for the chained GEMM.)

One pass = keys stationary (the dK/dV kernel), which then also forms dQ's contribution of its key block for every query tile;
without atomics those contributions go to one f32 slab per key workgroup and a second pass sums the slabs in fixed order.
At S = 40 962, H = 16, D = 64 a slab is S x H x D x 4 = 167.8 MB; the register file allows 64 keys per wave at one wave per SIMD
= 256 keys per workgroup = 160 slabs (26.8 GB), 512 keys per workgroup (80 slabs, 13.4 GB) would need two such workgroups' worth
of registers.  This tool times the two memory phases alone, with torch kernels at their best (a fill and a dim-0 sum), i.e. a
LOWER bound of what the real kernels would add -- to be set against the dQ kernel they replace (10.5 ms per layer) minus the
four extra MFMAs per tile pair the dK/dV kernel would take on (16 -> 20: ~ +4 ms at its present rate)."""
import torch

dev = "cuda"
S, H, D = 40962, 16, 64
row = S * H * D


def timed(fn, it=5):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it


for keys_per_wg in (256, 512):
    slabs = -(-S // keys_per_wg)
    buf = torch.empty((slabs, row), dtype=torch.float32, device=dev)
    gb = buf.numel() * 4 / 1e9
    t_w = timed(lambda: buf.fill_(1.0))
    out = torch.empty(row, dtype=torch.float32, device=dev)
    t_r = timed(lambda: torch.sum(buf, dim=0, out=out))
    print(f"{keys_per_wg} keys per workgroup: {slabs} slabs = {gb:.1f} GB of f32 partials per layer: written in {t_w:.2f} ms "
          f"({gb / t_w:.2f} TB/s), summed in fixed order in {t_r:.2f} ms ({gb / t_r:.2f} TB/s) -> {t_w + t_r:.1f} ms per layer on top of the "
          f"dK/dV kernel, against the 10.5 ms of the dQ kernel it replaces", flush=True)
    del buf, out
