#!/usr/bin/env python
"""CPU estimate of the k|v L2 miss traffic of the mesh edge kernel for a node order: every XCD walks its contiguous
eighth of the destinations in order; an LRU of R source rows stands for that XCD's L2.  Prints misses / compulsory."""
import sys
from collections import OrderedDict

import numpy as np
import torch

sys.path.insert(0, ".")
from anemoi_models_amd import runtime
from anemoi_models_amd.graphs.synthetic import build_graph


def misses(rowptr, col, n, cap, window=1):
    tot = 0
    for x in range(8):
        lru = OrderedDict()
        n0, n1 = n * x // 8, n * (x + 1) // 8
        for i in range(n0, n1):
            for j in col[rowptr[i]:rowptr[i + 1]]:
                if j in lru:
                    lru.move_to_end(j)
                else:
                    tot += 1
                    lru[j] = 1
                    if len(lru) > cap:
                        lru.popitem(last=False)
    return tot


def plan_for(order, ei, n):
    inv = runtime.inverse_permutation(order)
    e = torch.stack([inv[ei[0]], inv[ei[1]]])
    p = runtime.build_edge_plan(e, n, n)
    return p.rowptr.numpy(), p.col.numpy().tolist()


def main():
    g = build_graph(sys.argv[1] if len(sys.argv) > 1 else "n320_ico6")
    ei = g[("hidden", "to", "hidden")].edge_index
    n = g["hidden"].num_nodes
    lat, lon = g["hidden"].x[:, 0].double(), g["hidden"].x[:, 1].double()
    sincos = torch.stack([lat.sin(), lon.sin(), lat.cos(), lon.cos()], 1)
    orders = {"morton3d": runtime.locality_order(sincos), "natural": torch.arange(n)}
    for name, fn in runtime.__dict__.items():
        if name.startswith("locality_order_"):
            orders[name[15:]] = fn(sincos)
    for name, order in orders.items():
        rp, col = plan_for(order, ei, n)
        res = {cap: misses(rp, col, n, cap) / n for cap in (128, 256, 512, 1024)}
        print(name, {k: round(v, 2) for k, v in res.items()}, flush=True)


if __name__ == "__main__":
    main()
