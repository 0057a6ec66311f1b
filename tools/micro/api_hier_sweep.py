#!/usr/bin/env python
"""Constructor sweep of AnemoiModelEncProcDecHierarchical (see api_model_sweep.py): 2 / 3 hidden levels, level processing on /
off, channel and head counts, batch sizes, multistep inputs, f32 / bf16, eval / train."""
import itertools
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from anemoi_models_amd.graphs.synthetic import build_hierarchical_graph  # noqa: E402
from anemoi_models_amd.models import AnemoiModelEncProcDecHierarchical  # noqa: E402
from anemoi_models_amd.utils.indices import SimpleDataIndices  # noqa: E402
from anemoi_models_amd.utils.presets import hierarchical_model_config  # noqa: E402

dev = "cuda"
graphs = {2: build_hierarchical_graph("o32", (2, 1)), 3: build_hierarchical_graph("o32", (3, 2, 1))}
seen, ok, total = {}, 0, 0
for levels, level_process, (channels, heads), (b, multistep), level_layers, trainable in itertools.product(
        (2, 3), (True, False), ((64, 16), (128, 8), (64, 4)), ((1, 2), (2, 1), (1, 3)), (1, 2), (8, 0)):
    hidden = [f"hidden_{i + 1}" for i in range(levels)]
    graph = graphs[levels]
    for mode, train in itertools.product(("fp32", "bf16"), (False, True)):
        os.environ["ANEMOI_AMD_DTYPE"] = mode
        total += 1
        what = f"levels={levels} level_process={level_process} C={channels} H={heads} B={b} T={multistep} layers={level_layers} " \
               f"trainable={trainable} {mode} {'train' if train else 'eval'}"
        try:
            torch.manual_seed(1)
            idx = SimpleDataIndices(n_prognostic=10, n_forcing=2, n_diagnostic=1)
            cfg = hierarchical_model_config(channels, heads, hidden=hidden, level_layers=level_layers, level_process=level_process,
                                            multistep=multistep, trainable=trainable)
            model = AnemoiModelEncProcDecHierarchical(model_config=cfg, data_indices=idx, graph_data=graph).to(dev).train(train)
            x = torch.randn(b, multistep, 1, graph["data"].num_nodes, idx.num_input, device=dev)
            with torch.enable_grad() if train else torch.no_grad():
                y = model(x)
                assert y.shape == (b, 1, graph["data"].num_nodes, idx.num_output), y.shape
                if train:
                    y.float().square().mean().backward()
            assert bool(torch.isfinite(y).all())
            ok += 1
        except Exception as exc:  # noqa: BLE001
            msg = f"{type(exc).__name__}: {str(exc).splitlines()[0][:170] if str(exc) else ''}"
            seen.setdefault(msg, []).append(what)
print(f"{ok} of {total} combinations ran")
for msg, where in sorted(seen.items()):
    print(f"{msg}\n    {len(where)} cases, e.g. {where[0]} | {where[-1]}")
