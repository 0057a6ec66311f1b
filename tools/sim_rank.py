#!/usr/bin/env python
"""Compute-side step time of ONE rank of a P-way mesh partition, alone on one GPU (no communication: exchanges are
stubbed by distributed.partition.SimulatedRank, halo rows read as zeros -- the outputs are meaningless, the launches
and shapes are the real ones).  Estimates where strong scaling goes before an 8-GPU node is available:
   python tools/sim_rank.py [--workload cfg3] [--worlds 2,4,8] [--steps 10]
Prints, per world size, the slowest rank's ms/step, the host enqueue time and the ideal (1-GPU time / P)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from anemoi_models_amd.distributed.partition import SimulatedRank  # noqa: E402


def timed(fn, steps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    host = (time.perf_counter() - t0) / steps
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, host * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--worlds", default="2,4,8")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--ranks", default="all", help="'all' or a comma list")
    ap.add_argument("--detail", action="store_true")
    a = ap.parse_args()
    os.environ.setdefault("ANEMOI_AMD_DTYPE", "bf16")
    dev = torch.device("cuda", 0)
    model, graph, x, _ = bench.build(a.workload, dev)
    with torch.no_grad():
        t1, h1 = timed(lambda: model(x), a.steps)
        print(f"{a.workload} 1 GPU: {t1:.2f} ms / step (host enqueue {h1:.2f} ms)", flush=True)
        for world in [int(w) for w in a.worlds.split(",")]:
            ranks = range(world) if a.ranks == "all" else [int(r) for r in a.ranks.split(",")]
            res = []
            for r in ranks:
                grp = SimulatedRank(r, world)
                t, h = timed(lambda: model(x, grp), a.steps)
                res.append((t, h, r))
                if a.detail and r == ranks[0]:
                    bench.profile_pass(model, x, grp, "bf16", True, traffic_ok=False)
            worst = max(res)
            print(f"world {world}: slowest rank {worst[2]} {worst[0]:.2f} ms / step (host enqueue {worst[1]:.2f} ms), "
                  f"mean {sum(t for t, _, _ in res) / len(res):.2f}; ideal {t1 / world:.2f}; compute-side speed-up "
                  f"{t1 / worst[0]:.2f}x", flush=True)


if __name__ == "__main__":
    main()
