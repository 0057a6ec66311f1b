"""Worker of tests/test_gpu_parity.py::test_node_partitioned_forward_ranks_sharing_one_gpu.

usage: python _gpu_shared_ranks.py RANK WORLD PORT OUT GRAPH CHANNELS LAYERS HEADS DTYPE [train]
With ``train``: one node-partitioned training step (sharded forward with an autograd graph, backward with the reverse halo
all-to-all-v) -- the parameter gradients summed over the ranks must equal the single-device gradients.
Every rank runs the HIP kernels on cuda:0; the collectives are gloo, staged through host memory
(anemoi_models_amd/distributed/partition.py::_alltoallv).  Writes max |sharded - unsharded| to OUT.RANK.
"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    out, graph_name = sys.argv[4], sys.argv[5]
    channels, layers, heads, dtype = int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8]), sys.argv[9]
    os.environ["ANEMOI_AMD_DTYPE"] = dtype
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from anemoi_models_amd import _lib

    _lib.load()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from anemoi_models_amd.graphs.synthetic import build_graph
        from anemoi_models_amd.models import AnemoiModelEncProcDec
        from anemoi_models_amd.utils.indices import SimpleDataIndices
        from anemoi_models_amd.utils.presets import model_config

        device = torch.device("cuda", 0)
        graph = build_graph(graph_name)
        idx = SimpleDataIndices(n_prognostic=10, n_forcing=2, n_diagnostic=1)
        torch.manual_seed(1234)  # the same weights and input on every rank
        family = os.environ.get("ANEMOI_TEST_FAMILY", "GraphTransformer")  # GNN_all: GNN processor + GNN mappers
        proc_, map_ = ("GNN", "GNN") if family == "GNN_all" else (family, "GraphTransformer")
        model = AnemoiModelEncProcDec(model_config=model_config(proc_, channels, layers, heads, mappers=map_),
                                      data_indices=idx, graph_data=graph).to(device).eval()
        x = torch.randn(1, 2, 1, graph["data"].num_nodes, 12, device=device)
        with torch.no_grad():
            want = model(x)
            got = model(x, dist.group.WORLD)
            again = model(x, dist.group.WORLD)
        torch.cuda.synchronize()
        sp = [v for k, v in model._idx_cache.items() if k[0] == "shard_plan"][0]
        info = dict(err=float((got - want).abs().max()), rerun=float((again - got).abs().max()),
                    scale=float(want.abs().max()), own=sp.hi - sp.lo, finite=bool(torch.isfinite(got).all()))
        if family == "Transformer":
            # attention dropout (training mode, no autograd: the inference route with the rows <-> heads exchange): the same
            # CPU-generator draws on both routes -> rank 0's seed for all, global head index in the mask's hash -> the ranks
            # together drop what the unsharded attention drops
            rates = {m: m.dropout_p for m in model.modules() if hasattr(m, "dropout_p")}
            for m in rates:
                m.dropout_p = 0.3
            model.train()
            with torch.no_grad():
                torch.manual_seed(99)
                want_d = model(x)
                torch.manual_seed(99)
                got_d = model(x, dist.group.WORLD)
            info.update(drop_err=float((got_d - want_d).abs().max()), drop_acts=float((want_d - want).abs().max()))
            for m, rate in rates.items():
                m.dropout_p = rate
            model.eval()
        if len(sys.argv) > 10 and sys.argv[10] == "train":
            model.train()
            dy = torch.randn(want.shape, generator=torch.Generator().manual_seed(5)).to(device)
            y1 = model(x)  # single device, differentiable route
            y1.backward(dy)
            full = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
            model.zero_grad()
            y2 = model(x, dist.group.WORLD)  # node-partitioned, differentiable route
            y2.backward(dy)
            g_err, g_scale, missing = 0.0, 0.0, []
            for k, p in model.named_parameters():
                if k not in full:
                    continue
                part = (p.grad if p.grad is not None else torch.zeros_like(p)).cpu()
                dist.all_reduce(part)  # sum of the per-rank contributions (what DDP does over the model group)
                g_err = max(g_err, float((part - full[k].cpu()).abs().max()))
                g_scale = max(g_scale, float(full[k].abs().max()))
                if p.grad is None:
                    missing.append(k)
            info.update(train_out_err=float((y2.detach() - y1.detach()).abs().max()), grad_err=g_err, grad_scale=g_scale,
                        n_grads=len(full), requires_grad=bool(y2.requires_grad))
        torch.save(info, f"{out}.{rank}")
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
