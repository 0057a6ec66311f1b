"""Worker of tests/test_gpu_training.py::test_module_level_model_groups_ranks_sharing_one_gpu.

usage: python _gpu_shared_modules.py RANK WORLD PORT OUT
The reference's module-level calls with a model group (processors and mappers of all three families) on the HIP kernels,
forward and backward; the ranks are processes sharing cuda:0, the collectives gloo through host memory.  Writes the worst
deviation from the unsharded modules to OUT.RANK.
"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    os.environ["ANEMOI_AMD_DTYPE"] = "fp32"
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from anemoi_models_amd import _lib

    _lib.load()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from anemoi_models_amd.distributed.shapes import get_shape_shards
        from anemoi_models_amd.graphs.synthetic import build_graph
        from anemoi_models_amd.layers.mapper import GraphTransformerBackwardMapper, GraphTransformerForwardMapper
        from anemoi_models_amd.layers.processor import GNNProcessor, GraphTransformerProcessor, TransformerProcessor

        dev = torch.device("cuda", 0)
        g = dist.group.WORLD
        torch.manual_seed(7)  # same weights on every rank
        c, heads = 64, 4
        graph = build_graph("o32_ico2")
        n_h, n_d = graph["hidden"].num_nodes, graph["data"].num_nodes
        attrs = ["edge_length", "edge_dirs"]
        kw = dict(sub_graph=graph[("hidden", "to", "hidden")], sub_graph_edge_attributes=attrs, src_grid_size=n_h,
                  dst_grid_size=n_h)
        procs = {
            "gt": GraphTransformerProcessor(num_layers=2, trainable_size=2, num_channels=c, num_chunks=1, num_heads=heads,
                                            mlp_hidden_ratio=2, **kw),
            "gnn": GNNProcessor(num_layers=2, trainable_size=2, num_channels=c, num_chunks=1, mlp_extra_layers=0, **kw),
            "tfm": TransformerProcessor(num_layers=2, window_size=None, num_channels=c, num_chunks=1, num_heads=heads,
                                        mlp_hidden_ratio=2, dropout_p=0.0),
        }
        gen = torch.Generator().manual_seed(3)
        xh = torch.randn(n_h, c, generator=gen).to(dev)
        dy = torch.randn(n_h, c, generator=gen).to(dev)
        sh = get_shape_shards(xh, 0, g)
        rows = slice(sum(s[0] for s in sh[:rank]), sum(s[0] for s in sh[:rank + 1]))
        info = {}
        for name, proc in procs.items():
            proc = proc.to(dev).eval()
            with torch.no_grad():
                want = proc(xh, 1, [list(xh.shape)])
                got = proc(xh[rows].contiguous(), 1, sh, g)
            info[name + ".fwd"] = float((got - want[rows]).abs().max() / want.abs().max())
            # backward: the rank's share of the parameter gradients, summed over the ranks, equals the unsharded gradient
            proc.train()
            proc(xh, 1, [list(xh.shape)]).backward(dy)
            full = {k: p.grad.clone() for k, p in proc.named_parameters() if p.grad is not None}
            proc.zero_grad()
            proc(xh[rows].contiguous(), 1, sh, g).backward(dy[rows].contiguous())
            err = scale = 0.0
            for k, p in proc.named_parameters():
                if k not in full:
                    continue
                part = (p.grad if p.grad is not None else torch.zeros_like(p)).cpu()
                if "trainable" not in k:
                    dist.all_reduce(part)
                # (the trainable edge tensor is sharded by shard_tensor, whose backward GATHERS: every rank already holds
                #  its complete gradient -- the reference's semantics, distributed/graph.py:19-44; anemoi-training scales
                #  the other parameters' gradients by the group size instead of dividing this one)
                err = max(err, float((part - full[k].cpu()).abs().max()))
                scale = max(scale, float(full[k].abs().max()))
            info[name + ".grad"] = err / max(scale, 1e-30)
        # attention dropout across the group (reference layers/processor.py:99: dropout_p = 0.1 by default): rank 0's seed
        # for all, the global head index in the mask's hash -> the ranks together drop what the unsharded attention drops
        drop = TransformerProcessor(num_layers=2, window_size=None, num_channels=c, num_chunks=1, num_heads=heads,
                                    mlp_hidden_ratio=2, dropout_p=0.3).to(dev).train()
        with torch.no_grad():
            torch.manual_seed(99)  # the blocks draw their seeds from torch's CPU generator: the same draws on both routes
            want = drop(xh, 1, [list(xh.shape)])
            torch.manual_seed(99)
            got = drop(xh[rows].contiguous(), 1, sh, g)
            info["tfm_dropout.fwd"] = float((got - want[rows]).abs().max() / want.abs().max())
            info["tfm_dropout.acts"] = float((want - drop.eval()(xh, 1, [list(xh.shape)])).abs().max() > 0)  # dropout did act
        enc = GraphTransformerForwardMapper(in_channels_src=20, in_channels_dst=6, hidden_dim=c, trainable_size=2,
                                            num_heads=heads, mlp_hidden_ratio=2, sub_graph=graph[("data", "to", "hidden")],
                                            sub_graph_edge_attributes=attrs, src_grid_size=n_d, dst_grid_size=n_h).to(dev).eval()
        dec = GraphTransformerBackwardMapper(in_channels_src=c, in_channels_dst=20, hidden_dim=c, trainable_size=2,
                                             out_channels_dst=7, num_heads=heads, mlp_hidden_ratio=2,
                                             sub_graph=graph[("hidden", "to", "data")], sub_graph_edge_attributes=attrs,
                                             src_grid_size=n_h, dst_grid_size=n_d).to(dev).eval()
        xd, xh6 = torch.randn(n_d, 20, generator=gen).to(dev), torch.randn(n_h, 6, generator=gen).to(dev)
        sd_, sh6 = get_shape_shards(xd, 0, g), get_shape_shards(xh6, 0, g)
        with torch.no_grad():
            _, want = enc((xd, xh6), 1, ([list(xd.shape)], [list(xh6.shape)]))
            _, got = enc((xd, xh6), 1, (sd_, sh6), g)
            info["enc.fwd"] = float((got - want[rows]).abs().max() / want.abs().max())
            want = dec((xh, xd), 1, ([list(xh.shape)], [list(xd.shape)]))
            got = dec((xh[rows].contiguous(), xd), 1, (sh, sd_), g)
            info["dec.fwd"] = float((got - want).abs().max() / want.abs().max())
        torch.save(info, f"{out}.{rank}")
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
