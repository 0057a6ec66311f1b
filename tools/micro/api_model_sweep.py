#!/usr/bin/env python
"""Model-level companion of api_shape_sweep.py: AnemoiModelEncProcDec over processor / mapper families, channel counts, batch
and ensemble sizes, multistep inputs, variable counts, trainable-tensor sizes, processor chunks, attention windows -- eval and
training mode, f32 and bf16.  Prints every exception class once.  python tools/micro/api_model_sweep.py"""
import itertools
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from anemoi_models_amd.graphs.synthetic import build_graph  # noqa: E402
from anemoi_models_amd.models import AnemoiModelEncProcDec  # noqa: E402
from anemoi_models_amd.utils.indices import SimpleDataIndices  # noqa: E402
from anemoi_models_amd.utils.presets import model_config  # noqa: E402

dev = "cuda"
graph = build_graph("o32_ico2")
n_grid = graph["data"].num_nodes
seen, ok = {}, 0
cases = []
for proc, maps in (("GraphTransformer", "GraphTransformer"), ("GNN", "GraphTransformer"), ("GNN", "GNN"),
                   ("Transformer", "GraphTransformer"), ("GraphTransformer", "GNN"), ("Transformer", "GNN")):
    for channels, heads in ((64, 16), (128, 8), (192, 16), (256, 4), (64, 1)):
        for (b, ens), multistep, (n_prog, n_forc, n_diag), trainable, chunks in (
                ((1, 1), 2, (10, 2, 1), 8, 2), ((2, 1), 2, (10, 2, 1), 8, 2), ((1, 3), 1, (10, 2, 1), 8, 1),
                ((1, 1), 3, (5, 0, 0), 0, 4), ((3, 1), 1, (1, 0, 3), 3, 1), ((1, 1), 2, (26, 6, 1), 8, 3)):
            cases.append((proc, maps, channels, heads, b, ens, multistep, n_prog, n_forc, n_diag, trainable, chunks))
for case in cases:
    proc, maps, channels, heads, b, ens, multistep, n_prog, n_forc, n_diag, trainable, chunks = case
    for mode, train in itertools.product(("fp32", "bf16"), (False, True)):
        os.environ["ANEMOI_AMD_DTYPE"] = mode
        what = f"{proc} / {maps} mappers C={channels} H={heads} B={b} ens={ens} T={multistep} vars={n_prog}+{n_forc}+{n_diag} " \
               f"trainable={trainable} chunks={chunks} {mode} {'train' if train else 'eval'}"
        try:
            torch.manual_seed(1)
            idx = SimpleDataIndices(n_prognostic=n_prog, n_forcing=n_forc, n_diagnostic=n_diag)
            cfg = model_config(proc, channels, 4, heads, multistep=multistep, trainable=trainable, proc_chunks=chunks,
                               window_size=64, mappers=maps)
            if os.environ.get("SWEEP_MAPPER_CHUNKS"):  # (the mappers' own num_chunks: row chunks of their node MLPs)
                cfg["model"]["encoder"]["num_chunks"] = cfg["model"]["decoder"]["num_chunks"] = int(os.environ["SWEEP_MAPPER_CHUNKS"])
            model = AnemoiModelEncProcDec(model_config=cfg, data_indices=idx, graph_data=graph).to(dev).train(train)
            for m in model.modules():
                if hasattr(m, "dropout_p"):
                    m.dropout_p = 0.0
            x = torch.randn(b, multistep, ens, n_grid, idx.num_input, device=dev)
            with torch.enable_grad() if train else torch.no_grad():
                y = model(x)
                assert y.shape == (b, ens, n_grid, idx.num_output), y.shape
                if train:
                    y.float().square().mean().backward()
            assert bool(torch.isfinite(y).all())
            ok += 1
        except Exception as exc:  # noqa: BLE001
            msg = f"{type(exc).__name__}: {str(exc).splitlines()[0][:170] if str(exc) else ''}"
            seen.setdefault(msg, []).append(what)
print(f"{ok} of {4 * len(cases)} combinations ran")
for msg, where in sorted(seen.items()):
    print(f"{msg}\n    {len(where)} cases, e.g. {where[0]} | {where[-1]}")
