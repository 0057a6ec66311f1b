#!/usr/bin/env python
"""Why did test_config2_bf16_anchored_to_the_oracle_under_bf16_autocast move (round 6)?  The HIP bf16 encoder latent against
the f32 oracle's, with the oracle at two host thread counts: max / rms / tail of the error, where the maximum sits."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from oracle import reference_path as ref  # noqa: E402
from test_oracle_golden import graph_tensors  # noqa: E402
import test_gpu_baseline_sizes as T  # noqa: E402

os.environ["ANEMOI_AMD_DTYPE"] = "bf16"
for threads in (int(a) for a in sys.argv[1:] or ["16", "128"]):
    torch.set_num_threads(threads)
    model, x, want, graph, _ = T._make("GraphTransformer")
    sd = {k: (v.detach().float() if v.is_floating_point() else v.detach()).cpu() for k, v in model.state_dict().items()}
    kw = dict(num_heads=16, num_layers=16, num_chunks=2, prognostic_in=range(T.N_PROG), prognostic_out=range(T.N_PROG),
              return_stages=True)
    with torch.no_grad():
        want32, st32 = ref.model_forward(sd, graph_tensors(graph), x.cpu(), **kw)
    auto, st_auto = T.oracle_under_bf16_autocast(lambda: ref.model_forward(sd, graph_tensors(graph), x.cpu(), **kw))
    got, latent = bench.device_forward_with_latent(model, x)
    got2, latent2 = bench.device_forward_with_latent(model, x)
    lat, refl, autol = latent.float().cpu(), st32["x_latent"], st_auto["x_latent"].float()
    scale = float(refl.abs().max())
    for name, a in (("HIP bf16", lat), ("oracle under bf16 autocast", autol)):
        d = (a - refl).abs() / scale
        flat = d.flatten()
        top = torch.topk(flat, 5)
        print(f"threads {threads}: {name}: max {float(flat.max()):.4e} rms {float(flat.pow(2).mean().sqrt()):.3e} "
              f"p99.99 {float(torch.quantile(flat[:: 7], 0.9999)):.3e} top5 {[f'{v:.3e}' for v in top.values.tolist()]} at "
              f"{[(int(i) // d.shape[1], int(i) % d.shape[1]) for i in top.indices.tolist()]} "
              f"|ref| there {[f'{float(refl.flatten()[i]):.2f}' for i in top.indices.tolist()]} (scale {scale:.2f})", flush=True)
    # the suite runs an f32 forward of the SAME model object first (module fixture): does that change the bf16 result?
    os.environ["ANEMOI_AMD_DTYPE"] = "fp32"
    with torch.no_grad():
        model(x)
    os.environ["ANEMOI_AMD_DTYPE"] = "bf16"
    got3, latent3 = bench.device_forward_with_latent(model, x)
    d3 = (latent3.float().cpu() - refl).abs() / scale
    print(f"threads {threads}: bf16 after an f32 forward of the same model: latent identical to before: "
          f"{torch.equal(latent, latent3)}, max err {float(d3.max()):.4e}; prediction identical: {torch.equal(got, got3)}", flush=True)
    print(f"threads {threads}: two HIP runs identical: {torch.equal(latent, latent2)}; prediction HIP {T.rel_err(got, want32):.3e} "
          f"autocast {T.rel_err(auto, want32):.3e}; checksum latent {float(lat.double().sum()):.6f}", flush=True)
