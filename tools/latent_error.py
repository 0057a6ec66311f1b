#!/usr/bin/env python
"""Attribution of the bf16 encoder-latent error: the mesh latent of the N320 -> ico-6 encoder on the
HIP bf16 path against the f32 CPU oracle -- with every fold on, and with the embedding fold, the LayerNorm fold and the
lin_edge fold switched off in turn -- next to the yardstick: the ORACLE's own encoder under bf16 autocast (the
reference's production arithmetic, tests/test_gpu_baseline_sizes.py::_CudaAutocastPolicy) against the f32 oracle.

    python tools/latent_error.py [--workload cfg3]      (one GPU; ~1 min of host time at config 3)
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import torch  # noqa: E402

import bench  # noqa: E402
from oracle import reference_path as ref  # noqa: E402  (checker only)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg3")
    args = ap.parse_args()
    from test_gpu_baseline_sizes import oracle_under_bf16_autocast

    os.environ["ANEMOI_AMD_DTYPE"] = "bf16"
    model, graph, x, idx = bench.build(args.workload, torch.device("cuda", 0))
    sd = {k: (v.detach().float() if v.is_floating_point() else v.detach()).cpu() for k, v in model.state_dict().items()}
    data, hidden = model._graph_name_data, model._graph_name_hidden
    heads = model.processor.proc[0].blocks[0].num_heads
    xc = x.float().cpu()
    b, t, ens, g, v = xc.shape
    chunks = 8 if g > 200_000 else 1
    ea, ei = model.encoder.edge_attr.detach().float().cpu(), model.encoder.edge_index_base.cpu()

    def encoder():
        x_data = torch.cat((xc.permute(0, 2, 3, 1, 4).reshape(b * ens * g, t * v), ref.node_attributes(sd, data, b)), dim=-1)
        return ref.gt_forward_mapper(sd, "encoder", x_data, ref.node_attributes(sd, hidden, b), ea, ei, b, heads, "GELU", chunks)[1]

    with torch.no_grad():
        want = encoder()
    auto = oracle_under_bf16_autocast(encoder).float()

    def rel(a):
        return float((a.float().cpu() - want).abs().max() / want.abs().max())

    def rms(a):
        return float((a.float().cpu() - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt())

    rows = [("oracle under bf16 autocast (the reference's bf16-mixed arithmetic)", rel(auto), rms(auto))]
    for label, env in (("HIP bf16, all folds (shipped)", {}),
                       ("HIP bf16, ANEMOI_AMD_EMBED_FOLD=0", {"ANEMOI_AMD_EMBED_FOLD": "0"}),
                       ("HIP bf16, ANEMOI_AMD_LN_FOLD=0 (also turns the embedding fold off)", {"ANEMOI_AMD_LN_FOLD": "0"}),
                       ("HIP bf16, ANEMOI_AMD_EDGE_FOLD=0", {"ANEMOI_AMD_EDGE_FOLD": "0"}),
                       ("HIP f32 (exact-f32 MFMA)", {"ANEMOI_AMD_DTYPE": "fp32"})):
        saved = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            _, latent = bench.device_forward_with_latent(model, x)
            rows.append((label, rel(latent), rms(latent)))
        finally:
            for k, val in saved.items():
                if val is None:
                    del os.environ[k]
                else:
                    os.environ[k] = val
    print(f"encoder output (mesh latent [{want.shape[0]}, {want.shape[1]}]) against the f32 CPU oracle, {bench.WORKLOADS[args.workload][4]}")
    print(f"{'':72s} {'max|a-b|/max|b|':>16s} {'rms(a-b)/rms(b)':>16s}")
    for label, e, r in rows:
        print(f"{label:72s} {e:16.3e} {r:16.3e}")


if __name__ == "__main__":
    main()
