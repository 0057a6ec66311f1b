#!/usr/bin/env python
"""Which constructor arguments of the path's modules does this package refuse that the reference takes?  (round 6: the op
fuzzer found GraphTransformerConv raising at head sizes 5 / 12 / 20.)  Blocks of all three families over a grid of channel
counts, head counts and edge widths, eval and training mode, f32 and bf16: prints every exception once per (module, shape
class).  python tools/micro/api_shape_sweep.py"""
import itertools
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from anemoi_models_amd.layers.block import (GraphConvProcessorBlock, GraphTransformerMapperBlock,  # noqa: E402
                                            GraphTransformerProcessorBlock, TransformerProcessorBlock)

dev = "cuda"
g = torch.Generator().manual_seed(0)
n, e = 300, 2500
ei = torch.stack([torch.randint(0, n, (e,), generator=g), torch.randint(0, n, (e,), generator=g)]).to(dev)
seen, ok = {}, 0


def attempt(what, cls_key, fn):
    global ok
    try:
        out = fn()
        assert bool(torch.isfinite(out if isinstance(out, torch.Tensor) else out[0]).all())
        ok += 1
    except Exception as exc:  # noqa: BLE001
        msg = f"{type(exc).__name__}: {str(exc).splitlines()[0][:150]}"
        seen.setdefault((cls_key, msg), []).append(what)


for c, h, edge_dim in itertools.product([32, 64, 96, 128, 160, 192, 320, 512], [1, 2, 4, 8, 16], [3, 11, 39]):
    if c % h:
        continue
    for mode, train in itertools.product(("fp32", "bf16"), (False, True)):
        os.environ["ANEMOI_AMD_DTYPE"] = mode
        what = f"C={c} H={h} (D={c // h}) edge_dim={edge_dim} {mode} {'train' if train else 'eval'}"
        x = torch.randn(n, c, generator=g).to(dev).requires_grad_(train)
        ea = torch.randn(e, edge_dim, generator=g).to(dev)

        def run(blk, *args):
            blk = blk.to(dev).train(train)
            with torch.enable_grad() if train else torch.no_grad():
                out = blk(*args)
                y = out[0] if isinstance(out, tuple) else out
                y = y[1] if isinstance(y, tuple) else y  # (mapper block: ((x_src, x_dst), edge_attr))
                if train:
                    y.float().sum().backward()
            return y.detach()

        attempt(what, "GraphTransformerProcessorBlock",
                lambda: run(GraphTransformerProcessorBlock(c, 2 * c, c, edge_dim=edge_dim, num_heads=h), x, ea, ei, None, 1))
        attempt(what, "GraphTransformerMapperBlock",
                lambda: run(GraphTransformerMapperBlock(c, 2 * c, c, edge_dim=edge_dim, num_heads=h), (x, x), ea, ei,
                            (None, None, None), 1))
        if edge_dim == 3:
            attempt(what, "GraphConvProcessorBlock",
                    lambda: run(GraphConvProcessorBlock(c, c, num_chunks=1), x, torch.randn(e, c, generator=g).to(dev), ei,
                                (None, None, None)))
            attempt(what, "TransformerProcessorBlock",
                    lambda: run(TransformerProcessorBlock(c, 2 * c, h, "GELU", window_size=64), x, [list(x.shape)], 1))
print(f"{ok} combinations ran")
for (cls_key, msg), where in sorted(seen.items()):
    print(f"{cls_key}: {msg}\n    {len(where)} cases, e.g. {where[0]} | {where[-1]}")
