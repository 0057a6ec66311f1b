"""World-size-2 (and 3) gloo tests of the node-partitioned forward: partition plans, halo all-to-all-v, result gather.

The arithmetic is substituted by the oracle-backed CPU ops of tests/_cpu_ops.py (the HIP kernels need a GPU); what is
under test is the distributed host logic of anemoi_models_amd/distributed/partition.py.
"""

import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, ROOT


def _worker(rank, world, port, result_file, processor="GraphTransformer", dtype=None, heads=16, window=None):
    if dtype is not None:
        os.environ["ANEMOI_AMD_DTYPE"] = dtype
    if window is not None:  # flash-attn's sliding window semantics (reference layers/attention.py:96) instead of SDPA-global
        os.environ["ANEMOI_AMD_FLASH_WINDOW"] = "1"
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import numpy as np

        import _cpu_ops
        import anemoi_models_amd.ops as ops
        from anemoi_models_amd.graphs.synthetic import build_graph
        from anemoi_models_amd.models import AnemoiModelEncProcDec
        from anemoi_models_amd.utils.indices import SimpleDataIndices
        from anemoi_models_amd.utils.presets import model_config

        for name in ("layer_norm", "row_stats", "linear", "edge_attr_csr", "gt_edge_attention",
                     "gt_edge_attention_folded", "gather_add_act", "segment_sum", "mhsa", "assemble_nodes",
                     "prognostic_residual", "finalize_output", "convert_pad", "add", "act_forward"):
            setattr(ops, name, getattr(_cpu_ops, name))
        mappers = "GraphTransformer"
        if processor == "GNN_all":  # GNN processor AND GNN mappers
            processor, mappers = "GNN", "GNN"
        fname = {"GraphTransformer": "cfg1_gt.npz", "GNN": "cfg1_gnn.npz", "Transformer": "cfg1_tfm.npz"}[processor]
        if mappers == "GNN":
            fname = "cfg1_gnn_all.npz"
        with np.load(os.path.join(GOLDEN, fname)) as z:
            gold = {k: torch.from_numpy(z[k]) for k in z.files}
        graph = build_graph("o32_ico2")
        idx = SimpleDataIndices(n_prognostic=10, n_forcing=2, n_diagnostic=1)
        cfg = model_config(processor, 64, 4, heads, mappers=mappers) if window is None else model_config(
            processor, 64, 4, heads, mappers=mappers, window_size=window)
        model = AnemoiModelEncProcDec(model_config=cfg, data_indices=idx, graph_data=graph)
        model.load_state_dict({k[3:]: v for k, v in gold.items() if k.startswith("sd.")})
        model.eval()
        with torch.no_grad():
            y = model(gold["x"], dist.group.WORLD)
            # the golden output belongs to 16 heads and global attention; anything else is compared with the unsharded forward
            want = gold["y"] if heads == 16 and window is None else model(gold["x"])
        err = float((y - want).abs().max())
        # every rank must hold the full output; halo / partition sanity
        sp = [v for k, v in model._idx_cache.items() if k[0] == "shard_plan"][0]
        n_mesh = graph["hidden"].num_nodes
        info = dict(err=err, rank=rank, own=sp.hi - sp.lo,
                    halo=sp.proc.halo.n_recv if sp.proc.halo is not None else 1, dec_rows=int(sp.dec_dst_ids.numel()),
                    dec_halo=sp.dec.halo.n_recv, n_mesh=n_mesh, enc_src=int(sp.enc_src_ids.numel()))
        torch.save(info, f"{result_file}.{rank}")
    finally:
        dist.destroy_process_group()


def test_sharded_forward_bf16_with_layer_norm_fold(tmp_path):
    """The halo path of the blocks with the LayerNorm fold active (bf16 only): world 2 against the unsharded forward."""
    port = 29900 + (os.getpid() % 200)
    result = str(tmp_path / "res")
    # 4 heads of 16 channels: the folded edge kernel (needed by the partitioned path) wants >= 8 bf16 channels per head
    mp.spawn(_worker, args=(2, port, result, "GraphTransformer", "bf16", 4), nprocs=2, join=True)
    for r in range(2):
        assert torch.load(f"{result}.{r}")["err"] < 0.1  # sharded vs unsharded, both bf16 end to end (output scale ~ 5)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_forward_matches_golden(world, tmp_path):
    port = 29600 + world + (os.getpid() % 200)
    result = str(tmp_path / "res")
    mp.spawn(_worker, args=(world, port, result), nprocs=world, join=True)
    infos = [torch.load(f"{result}.{r}") for r in range(world)]
    for i in infos:
        assert i["err"] < 1e-4, i  # same function as the single-device forward (golden from the real reference)
    assert sum(i["own"] for i in infos) == infos[0]["n_mesh"]
    assert sum(i["dec_rows"] for i in infos) == 5248
    assert all(i["halo"] > 0 for i in infos)


def test_split_bounds_match_tensor_split():
    from anemoi_models_amd.distributed.shapes import split_bounds

    for n, p in [(10, 3), (162, 8), (40962, 8), (7, 7), (5, 8)]:
        sizes = [t.shape[0] for t in torch.arange(n).tensor_split(p)]
        b = split_bounds(n, p)
        assert [b[i + 1] - b[i] for i in range(p)] == sizes


@pytest.mark.parametrize("processor", ["GNN", "Transformer", "GNN_all"])
def test_sharded_forward_other_processors(processor, tmp_path):
    world = 2
    port = 29700 + (os.getpid() % 200) + {"GNN": 7, "Transformer": 13, "GNN_all": 19}[processor]
    result = str(tmp_path / "res")
    mp.spawn(_worker, args=(world, port, result, processor), nprocs=world, join=True)
    for r in range(world):
        assert torch.load(f"{result}.{r}")["err"] < 1e-4


def test_sharded_forward_sliding_window_attention(tmp_path):
    """Transformer processor with flash-attn window semantics, node-partitioned: the heads <-> rows exchange delivers the
    full sequence in the internal mesh order, the window slides over the external order (permutation around the
    attention) -- sharded == unsharded for a window of 12 of the 162 mesh nodes."""
    world = 2
    port = 29650 + (os.getpid() % 200)
    result = str(tmp_path / "res")
    mp.spawn(_worker, args=(world, port, result, "Transformer", None, 16, 12), nprocs=world, join=True)
    for r in range(world):
        assert torch.load(f"{result}.{r}")["err"] < 1e-4


def _rollout_worker(rank, world, port, result_file):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import numpy as np

        import _cpu_ops
        import anemoi_models_amd.ops as ops
        from anemoi_models_amd.graphs.synthetic import build_graph
        from test_host_logic import build_interface

        for name in ("layer_norm", "row_stats", "linear", "edge_attr_csr", "gt_edge_attention", "gt_edge_attention_folded",
                     "gather_add_act", "segment_sum", "mhsa", "assemble_nodes",
                     "prognostic_residual", "finalize_output", "bound_output", "advance_input", "convert_pad", "add",
                     "act_forward"):
            setattr(ops, name, getattr(_cpu_ops, name))
        with np.load(os.path.join(GOLDEN, "interface_gt.npz")) as z:
            gold = {k: torch.from_numpy(z[k]) for k in z.files}
        iface = build_interface(build_graph("o32_ico2"), gold)
        iface.load_state_dict({k[3:]: v for k, v in gold.items() if k.startswith("sd.")})
        iface.eval()
        batch, forc = gold["batch"], gold["rollout_forcings"]
        every = iface.rollout(batch, 3, forc, dist.group.WORLD)                  # all-gather at every step
        last = iface.rollout(batch, 3, forc, dist.group.WORLD, gather="last")    # state kept sharded, gather at the end
        sp = [v for k, v in iface.model._idx_cache.items() if k[0] == "shard_plan"][0]
        torch.save(dict(err_golden=float((every - gold["rollout_y"]).abs().max()), shape_last=tuple(last.shape),
                        err_last=float((last[0] - every[-1]).abs().max()), scale=float(every.abs().max()),
                        grid_halo=int(sp.grid_halo_ids.numel()), grid_sent=sum(sp.grid_halo.send_splits),
                        grid=int(batch.shape[2])), f"{result_file}.{rank}")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_rollout_with_the_state_kept_sharded(world, tmp_path):
    """``rollout(..., gather="last")`` advances the state on each rank's own grid rows plus a small
    grid halo (one all-to-all-v of predictions per step) and all-gathers only the last forecast: same result as the
    per-step all-gather route, which itself reproduces the reference-chained golden rollout."""
    port = 29800 + world + (os.getpid() % 150)
    result = str(tmp_path / "res")
    mp.spawn(_rollout_worker, args=(world, port, result), nprocs=world, join=True)
    infos = [torch.load(f"{result}.{r}") for r in range(world)]
    for i in infos:
        assert i["err_golden"] < 2e-3 * i["scale"], i
        assert i["shape_last"][0] == 1 and i["err_last"] < 1e-5 * i["scale"], i
        assert 0 < i["grid_halo"] < i["grid"] // 2  # a boundary strip, not the grid
    assert sum(i["grid_halo"] for i in infos) == sum(i["grid_sent"] for i in infos)


def _collectives_worker(rank, world, port, result_file):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from anemoi_models_amd.distributed.graph import gather_tensor, reduce_shard_tensor, reduce_tensor, shard_tensor
        from anemoi_models_amd.distributed.graph import sync_tensor
        from anemoi_models_amd.distributed.transformer import shard_heads, shard_sequence

        g = dist.group.WORLD
        rows = [3, 5, 4][:world]  # unequal shards
        shapes = [[n, 6] for n in rows]

        def rnd(seed, *shape):
            return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))

        xs = [rnd(10 + r, rows[r], 6) for r in range(world)]          # what every rank holds
        ws = [rnd(20 + r, sum(rows), 6) for r in range(world)]        # every rank's weights on a gathered tensor
        vs = [rnd(30 + r, rows[r], 6) for r in range(world)]          # ... on its own shard
        full = torch.cat(xs, 0)
        lo = sum(rows[:rank])
        own = slice(lo, lo + rows[rank])
        errs = {}

        def check(name, got, want):
            errs[name] = float((got - want).abs().max()) if got.shape == want.shape else float("inf")

        # gather forward / take backward
        x = xs[rank].clone().requires_grad_()
        y = gather_tensor(x, 0, shapes, g)
        check("gather.fwd", y.detach(), full)
        (y * ws[rank]).sum().backward()
        check("gather.bwd", x.grad, ws[rank][own])
        # take forward / gather backward (and the variant that fills only the own slot)
        x = full.clone().requires_grad_()
        y = shard_tensor(x, 0, shapes, g)
        check("shard.fwd", y.detach(), xs[rank])
        (y * vs[rank]).sum().backward()
        check("shard.bwd", x.grad, torch.cat(vs, 0))
        x = full.clone().requires_grad_()
        (shard_tensor(x, 0, shapes, g, gather_in_backward=False) * vs[rank]).sum().backward()
        want = torch.zeros_like(full)
        want[own] = vs[rank]
        check("shard.bwd_local", x.grad, want)
        # column shards of a 3-D tensor: dim = -1 style use (dim index 2)
        cols = [2, 4, 1][:world]
        cshapes = [[2, 3, c] for c in cols]
        parts = [rnd(40 + r, 2, 3, cols[r]) for r in range(world)]
        y = gather_tensor(parts[rank].clone(), 2, cshapes, g)
        check("gather.dim2", y, torch.cat(parts, 2))
        # all-reduce forward, identity backward (bf16 input: f32 accumulation, result rounded once)
        same = [rnd(50 + r, 4, 6) for r in range(world)]
        x = same[rank].clone().requires_grad_()
        y = reduce_tensor(x, g)
        check("reduce.fwd", y.detach(), sum(same))
        (y * 2.0).sum().backward()
        check("reduce.bwd", x.grad, torch.full_like(x, 2.0))
        yb = reduce_tensor(same[rank].bfloat16(), g)
        check("reduce.bf16", yb.float(), sum(s.bfloat16().float() for s in same).bfloat16().float())
        # gather forward, all-reduce + split backward
        x = xs[rank].clone().requires_grad_()
        (sync_tensor(x, 0, shapes, g) * ws[rank]).sum().backward()
        check("sync.bwd", x.grad, sum(ws)[own])
        # all-reduce + split forward, gather backward
        fulls = [rnd(60 + r, sum(rows), 6) for r in range(world)]
        x = fulls[rank].clone().requires_grad_()
        y = reduce_shard_tensor(x, 0, shapes, g)
        check("reduce_shard.fwd", y.detach(), sum(fulls)[own])
        (y * vs[rank]).sum().backward()
        check("reduce_shard.bwd", x.grad, torch.cat(vs, 0))
        # heads <-> sequence (batch 2, heads 2 * world, channels 3; sequence shards as tensor_split cuts them)
        heads, n_total = 2 * world, 4 * world + 1
        seq = [len(c) for c in torch.tensor_split(torch.arange(n_total), world)]
        sshapes = [[n, 3] for n in seq]
        qs = [rnd(70 + r, 2, heads, seq[r], 3) for r in range(world)]
        q_full = torch.cat(qs, 2)
        x = qs[rank].clone().requires_grad_()
        y = shard_heads(x, sshapes, g)
        check("heads.fwd", y.detach(), q_full[:, 2 * rank: 2 * rank + 2])
        z = shard_sequence(y, sshapes, g)
        check("heads.roundtrip", z.detach(), qs[rank])
        us = [rnd(80 + r, 2, 2, n_total, 3) for r in range(world)]  # weights on every rank's head shard
        (y * us[rank]).sum().backward()
        s0 = sum(seq[:rank])
        check("heads.bwd", x.grad, torch.cat(us, 1)[:, :, s0: s0 + seq[rank]])
        # no group: identity, the input itself
        t = rnd(1, 3, 3)
        errs["identity"] = 0.0 if (shard_tensor(t, 0, shapes, None) is t and gather_tensor(t, 0, shapes, None) is t
                                   and shard_heads(t, sshapes, None) is t) else 1.0
        torch.save(errs, f"{result_file}.{rank}")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_reference_collective_operators(world, tmp_path):
    """shard / gather / reduce / sync / reduce_shard / shard_heads / shard_sequence (reference distributed/graph.py:19-137,
    distributed/transformer.py:85-130): forward values and gradients on unequal shards against what one process computes
    from all ranks' tensors."""
    port = 29300 + (os.getpid() % 200) + world
    result = str(tmp_path / "res")
    mp.spawn(_collectives_worker, args=(world, port, result), nprocs=world, join=True)
    for r in range(world):
        errs = torch.load(f"{result}.{r}")
        assert len(errs) == 16, sorted(errs)
        for name, e in errs.items():
            assert e < 1e-6, (r, name, e)


def _sharded_attention_worker(rank, world, port, result_file):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import _cpu_ops
        import anemoi_models_amd.ops as ops
        from anemoi_models_amd.distributed.shapes import get_shape_shards
        from anemoi_models_amd.layers.attention import MultiHeadSelfAttention

        for name in ("linear", "mhsa", "convert_pad"):
            setattr(ops, name, getattr(_cpu_ops, name))
        torch.manual_seed(0)
        att = MultiHeadSelfAttention(num_heads=4, embed_dim=64, window_size=None, dropout_p=0.0).eval()
        n = 37
        x = torch.randn(n, 64, generator=torch.Generator().manual_seed(1))
        shapes = get_shape_shards(x, 0, dist.group.WORLD)
        lo = sum(s[0] for s in shapes[:rank])
        own = slice(lo, lo + shapes[rank][0])
        with torch.no_grad():
            want = att(x, [list(x.shape)], 1)
            got = att(x[own].contiguous(), shapes, 1, dist.group.WORLD)
        err = float((got - want[own]).abs().max())
        # the whole block (LayerNorms, MLP and residuals are row-local)
        from anemoi_models_amd.layers.block import TransformerProcessorBlock

        for name in ("layer_norm", "add", "act_forward"):
            setattr(ops, name, getattr(_cpu_ops, name))
        blk = TransformerProcessorBlock(64, 128, 4, "GELU", window_size=None, dropout_p=0.0).eval()
        with torch.no_grad():
            want = blk(x, [list(x.shape)], 1)
            got = blk(x[own].contiguous(), shapes, 1, dist.group.WORLD)
        err = max(err, float((got - want[own]).abs().max()))
        # ... and the processor (reference layers/processor.py:103-137: shard shapes + group handed to every block)
        from anemoi_models_amd.layers.processor import TransformerProcessor

        proc = TransformerProcessor(num_layers=2, window_size=None, num_channels=64, num_chunks=1, num_heads=4,
                                    mlp_hidden_ratio=2, dropout_p=0.0).eval()
        with torch.no_grad():
            want = proc(x, 1, [list(x.shape)])
            got = proc(x[own].contiguous(), 1, shapes, dist.group.WORLD)
        err = max(err, float((got - want[own]).abs().max()))
        # the graph blocks' resharding helpers (reference layers/block.py:366-414): q on the destination shards, k / v on
        # the source shards, the projected edge features on the edge shards -> all rows, this rank's heads; and back
        from anemoi_models_amd.layers.block import GraphTransformerProcessorBlock

        gblk = GraphTransformerProcessorBlock(64, 128, 64, edge_dim=3, num_heads=4)
        n_src, n_e = 29, 41
        full = {k: torch.randn(m, 64, generator=torch.Generator().manual_seed(s_))
                for k, m, s_ in (("q", n, 5), ("k", n_src, 6), ("v", n_src, 7), ("e", n_e, 8))}
        shp = {k: get_shape_shards(t, 0, dist.group.WORLD) for k, t in full.items()}
        mine = {k: t[sum(r[0] for r in shp[k][:rank]): sum(r[0] for r in shp[k][:rank + 1])].contiguous()
                for k, t in full.items()}
        q, k, v, e = gblk.shard_qkve_heads(mine["q"], mine["k"], mine["v"], mine["e"], (shp["k"], shp["q"], shp["e"]), 1,
                                           dist.group.WORLD)
        hs = slice(2 * rank, 2 * rank + 2)  # 4 heads over 2 ranks
        for got_t, name in ((q, "q"), (k, "k"), (v, "v"), (e, "e")):
            want_t = full[name].view(-1, 4, 16)[:, hs]
            err = max(err, float((got_t - want_t).abs().max()) if got_t.shape == want_t.shape else 1e9)
        back = gblk.shard_output_seq(q, (shp["k"], shp["q"], shp["e"]), 1, dist.group.WORLD)
        err = max(err, float((back - mine["q"]).abs().max()) if back.shape == mine["q"].shape else 1e9)
        torch.save(err, f"{result_file}.{rank}")
    finally:
        dist.destroy_process_group()


def test_module_level_sequence_sharded_attention(tmp_path):
    """MultiHeadSelfAttention.forward(x_shard, shapes, 1, group) as the reference shards it (layers/attention.py:72-112:
    shard_heads -> attention on the local heads -> shard_sequence), world 2 with unequal row shards, against the
    unsharded module."""
    port = 29500 + (os.getpid() % 200)
    result = str(tmp_path / "res")
    mp.spawn(_sharded_attention_worker, args=(2, port, result), nprocs=2, join=True)
    for r in range(2):
        assert torch.load(f"{result}.{r}") < 1e-5


def _sharded_gt_worker(rank, world, port, result_file):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import _cpu_ops
        import anemoi_models_amd.ops as ops
        from anemoi_models_amd.distributed.shapes import get_shape_shards
        from anemoi_models_amd.graphs.synthetic import build_graph
        from anemoi_models_amd.layers.block import GraphTransformerProcessorBlock
        from anemoi_models_amd.layers.processor import GraphTransformerProcessor

        for name in ("layer_norm", "layer_norm_with_stats", "row_stats", "linear", "linear_dual", "edge_attr_csr",
                     "gt_edge_attention", "gt_edge_attention_folded", "gt_conv", "convert_pad", "add", "act_forward"):
            setattr(ops, name, getattr(_cpu_ops, name))
        g = dist.group.WORLD
        torch.manual_seed(0)
        n, n_e, c, heads, edge_dim = 45, 160, 64, 4, 5
        gen = torch.Generator().manual_seed(3)
        x = torch.randn(n, c, generator=gen)
        ea = torch.randn(n_e, edge_dim, generator=gen)
        ei = torch.stack([torch.randint(0, n, (n_e,), generator=gen), torch.randint(0, n, (n_e,), generator=gen)])
        blk = GraphTransformerProcessorBlock(c, 2 * c, c, edge_dim=edge_dim, num_heads=heads).eval()
        sx, se = get_shape_shards(x, 0, g), get_shape_shards(ea, 0, g)
        rows = slice(sum(s[0] for s in sx[:rank]), sum(s[0] for s in sx[:rank + 1]))
        edges = slice(sum(s[0] for s in se[:rank]), sum(s[0] for s in se[:rank + 1]))
        with torch.no_grad():
            want, _ = blk(x, ea, ei, (None, None, None), 1)
            got, _ = blk(x[rows].contiguous(), ea[edges].contiguous(), ei, (sx, sx, se), 1, g, size=(n, n))
        err = float((got - want[rows]).abs().max())
        # the mapper block with update_src_nodes (reference layers/block.py:540-546: the source MLP runs on the source rows a
        # rank holds): 30 sources -> 45 destinations
        from anemoi_models_amd.layers.block import GraphTransformerMapperBlock

        n_s = 30
        xs_ = torch.randn(n_s, c, generator=gen)
        ei2 = torch.stack([torch.randint(0, n_s, (n_e,), generator=gen), torch.randint(0, n, (n_e,), generator=gen)])
        mblk = GraphTransformerMapperBlock(c, 2 * c, c, edge_dim=edge_dim, num_heads=heads, update_src_nodes=True).eval()
        ss = get_shape_shards(xs_, 0, g)
        srows = slice(sum(s[0] for s in ss[:rank]), sum(s[0] for s in ss[:rank + 1]))
        with torch.no_grad():
            (want_s, want_d), _ = mblk((xs_, x), ea, ei2, (None, None, None), 1, size=(n_s, n))
            (got_s, got_d), _ = mblk((xs_[srows].contiguous(), x[rows].contiguous()), ea[edges].contiguous(), ei2,
                                     (ss, sx, se), 1, g, size=(n_s, n))
        err = max(err, float((got_d - want_d[rows]).abs().max()), float((got_s - want_s[srows]).abs().max()),
                  0.0 if float((want_s - xs_).abs().max()) > 1e-3 else 1e9)  # (the source rows really were updated)
        # the processor on a real sub-graph (its own edge buffers and trainable edge tensor)
        graph = build_graph("o32_ico2")
        sub = graph[("hidden", "to", "hidden")]
        n_h = graph["hidden"].num_nodes
        proc = GraphTransformerProcessor(num_layers=2, trainable_size=2, num_channels=c, num_chunks=1, num_heads=heads,
                                         mlp_hidden_ratio=2, sub_graph=sub, sub_graph_edge_attributes=["edge_length", "edge_dirs"],
                                         src_grid_size=n_h, dst_grid_size=n_h).eval()
        with torch.no_grad():
            proc.trainable.trainable.normal_(0.0, 0.3, generator=torch.Generator().manual_seed(9))
        xh = torch.randn(n_h, c, generator=gen)
        sh = get_shape_shards(xh, 0, g)
        hrows = slice(sum(s[0] for s in sh[:rank]), sum(s[0] for s in sh[:rank + 1]))
        with torch.no_grad():
            want = proc(xh, 1, [list(xh.shape)])
            got = proc(xh[hrows].contiguous(), 1, sh, g)
        err = max(err, float((got - want[hrows]).abs().max()))
        # the mappers (reference layers/mapper.py:239-272, 275-418): the forward mapper takes the FULL node tensors and
        # shards them itself, the backward mapper takes a source shard + the full destination and gathers its output
        from anemoi_models_amd.layers.mapper import GraphTransformerBackwardMapper, GraphTransformerForwardMapper

        n_d = graph["data"].num_nodes
        enc = GraphTransformerForwardMapper(in_channels_src=20, in_channels_dst=6, hidden_dim=c, trainable_size=2, num_heads=heads,
                                            mlp_hidden_ratio=2, sub_graph=graph[("data", "to", "hidden")],
                                            sub_graph_edge_attributes=["edge_length", "edge_dirs"], src_grid_size=n_d,
                                            dst_grid_size=n_h).eval()
        dec = GraphTransformerBackwardMapper(in_channels_src=c, in_channels_dst=20, hidden_dim=c, trainable_size=2,
                                             out_channels_dst=7, num_heads=heads, mlp_hidden_ratio=2,
                                             sub_graph=graph[("hidden", "to", "data")],
                                             sub_graph_edge_attributes=["edge_length", "edge_dirs"], src_grid_size=n_h,
                                             dst_grid_size=n_d).eval()
        xd, xh6 = torch.randn(n_d, 20, generator=gen), torch.randn(n_h, 6, generator=gen)
        sd_, sh6 = get_shape_shards(xd, 0, g), get_shape_shards(xh6, 0, g)
        with torch.no_grad():
            _, want = enc((xd, xh6), 1, ([list(xd.shape)], [list(xh6.shape)]))
            raw, got = enc((xd, xh6), 1, (sd_, sh6), g)
        err = max(err, float((got - want[hrows]).abs().max()), 0.0 if raw is xd else 1e9)
        xs = torch.randn(n_h, c, generator=gen)
        with torch.no_grad():
            want = dec((xs, xd), 1, ([list(xs.shape)], [list(xd.shape)]))
            got = dec((xs[hrows].contiguous(), xd), 1, (sh, sd_), g)  # gathered: every rank holds all destination rows
        err = max(err, float((got - want).abs().max()) if got.shape == want.shape else 1e9)
        torch.save(err, f"{result_file}.{rank}")
    finally:
        dist.destroy_process_group()


def test_module_level_sharded_graph_transformer_block_and_processor(tmp_path):
    """GraphTransformerProcessorBlock / GraphTransformerProcessor / GraphTransformerForwardMapper / ...BackwardMapper called
    as the reference calls them across a model group (layers/block.py:479-635, layers/processor.py:317-343, layers/mapper.py:
    239-418: row shards of the nodes, edge shards of the attributes, the whole edge index, heads exchanged around the
    conv): world 2 against the unsharded modules."""
    port = 29700 + (os.getpid() % 200)
    result = str(tmp_path / "res")
    mp.spawn(_sharded_gt_worker, args=(2, port, result), nprocs=2, join=True)
    for r in range(2):
        assert torch.load(f"{result}.{r}") < 2e-5


def _sharded_gnn_worker(rank, world, port, result_file):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import _cpu_ops
        import anemoi_models_amd.ops as ops
        from anemoi_models_amd.distributed.shapes import get_shape_shards
        from anemoi_models_amd.graphs.synthetic import build_graph
        from anemoi_models_amd.layers.mapper import GNNBackwardMapper, GNNForwardMapper
        from anemoi_models_amd.layers.processor import GNNProcessor

        for name in ("layer_norm", "layer_norm_with_stats", "row_stats", "linear", "linear_dual", "edge_attr_csr",
                     "gather_add_act", "segment_sum", "convert_pad", "add", "act_forward"):
            setattr(ops, name, getattr(_cpu_ops, name))
        g = dist.group.WORLD
        torch.manual_seed(0)
        gen = torch.Generator().manual_seed(4)
        c = 64
        graph = build_graph("o32_ico2")
        n_h, n_d = graph["hidden"].num_nodes, graph["data"].num_nodes
        attrs = ["edge_length", "edge_dirs"]
        proc = GNNProcessor(num_layers=2, trainable_size=2, num_channels=c, num_chunks=1, mlp_extra_layers=0,
                            sub_graph=graph[("hidden", "to", "hidden")], sub_graph_edge_attributes=attrs, src_grid_size=n_h,
                            dst_grid_size=n_h).eval()
        with torch.no_grad():
            proc.trainable.trainable.normal_(0.0, 0.3, generator=torch.Generator().manual_seed(9))
        xh = torch.randn(n_h, c, generator=gen)
        sh = get_shape_shards(xh, 0, g)
        hrows = slice(sum(s[0] for s in sh[:rank]), sum(s[0] for s in sh[:rank + 1]))
        with torch.no_grad():
            want = proc(xh, 1, [list(xh.shape)])
            got = proc(xh[hrows].contiguous(), 1, sh, g)
        err = float((got - want[hrows]).abs().max())
        enc = GNNForwardMapper(in_channels_src=20, in_channels_dst=6, hidden_dim=c, trainable_size=2, mlp_extra_layers=0,
                               sub_graph=graph[("data", "to", "hidden")], sub_graph_edge_attributes=attrs, src_grid_size=n_d,
                               dst_grid_size=n_h).eval()
        dec = GNNBackwardMapper(in_channels_src=c, in_channels_dst=c, hidden_dim=c, trainable_size=2, out_channels_dst=7,
                                mlp_extra_layers=0, sub_graph=graph[("hidden", "to", "data")], sub_graph_edge_attributes=attrs,
                                src_grid_size=n_h, dst_grid_size=n_d).eval()
        xd, xh6 = torch.randn(n_d, 20, generator=gen), torch.randn(n_h, 6, generator=gen)
        sd_, sh6 = get_shape_shards(xd, 0, g), get_shape_shards(xh6, 0, g)
        drows = slice(sum(s[0] for s in sd_[:rank]), sum(s[0] for s in sd_[:rank + 1]))
        with torch.no_grad():
            want_src, want_dst = enc((xd, xh6), 1, ([list(xd.shape)], [list(xh6.shape)]))
            got_src, got_dst = enc((xd, xh6), 1, (sd_, sh6), g)  # full inputs in, row shards of both node sets out
        err = max(err, float((got_dst - want_dst[hrows]).abs().max()), float((got_src - want_src[drows]).abs().max()))
        xs, xdl = torch.randn(n_h, c, generator=gen), torch.randn(n_d, c, generator=gen)
        sdl = get_shape_shards(xdl, 0, g)
        with torch.no_grad():
            want = dec((xs, xdl), 1, ([list(xs.shape)], [list(xdl.shape)]))
            got = dec((xs[hrows].contiguous(), xdl[drows].contiguous()), 1, (sh, sdl), g)  # gathered output
        err = max(err, float((got - want).abs().max()) if got.shape == want.shape else 1e9)
        torch.save(err, f"{result_file}.{rank}")
    finally:
        dist.destroy_process_group()


def test_module_level_sharded_gnn_processor_and_mappers(tmp_path):
    """GNNProcessor / GNNForwardMapper / GNNBackwardMapper called as the reference calls them across a model group
    (layers/processor.py:228-250, layers/mapper.py:485-522, 600-705, layers/block.py:193-286: 1-hop edge shards,
    ``sync_tensor`` of the nodes, conv, ``shard_tensor`` of the sums): world 2 against the unsharded modules."""
    port = 29100 + (os.getpid() % 200)
    result = str(tmp_path / "res")
    mp.spawn(_sharded_gnn_worker, args=(2, port, result), nprocs=2, join=True)
    for r in range(2):
        assert torch.load(f"{result}.{r}") < 2e-5


def _empty_halo_worker(rank, world, port, result_file):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from anemoi_models_amd.distributed import partition

        calls = []
        real = partition._alltoallv

        def counting(out, inp, out_splits, in_splits, group, async_op=False):
            calls.append((tuple(out_splits), tuple(in_splits)))
            return real(out, inp, out_splits, in_splits, group, async_op=async_op)

        partition._alltoallv = counting
        # ranks 0 and 1 need two / one of each other's rows; the LAST rank owns no boundary rows at all (empty halo)
        n_own, width = 4, 3
        send = {0: ([1, 3], [0, 2, 0]), 1: ([2], [1, 0, 0])}.get(rank, ([], [0] * world))
        recv = {0: [0, 1, 0], 1: [2, 0, 0]}.get(rank, [0] * world)
        halo = partition.HaloExchange(torch.tensor(send[0], dtype=torch.int64), send[1][:world], recv[:world],
                                      dist.group.WORLD)
        rows = (torch.arange(n_own * width, dtype=torch.float32).view(n_own, width) + 100 * rank).requires_grad_()
        full = partition._HaloRows.apply(rows, halo)
        fwd_calls = len(calls)
        weight = torch.arange(full.numel(), dtype=torch.float32).view_as(full) + 1 + rank
        (full * weight).sum().backward()
        # what one process computes: d rows = own-row weights + the weights of every halo copy held by the peers
        want = weight[:n_own].clone()
        if rank == 0:  # rank 1 holds copies of rows 1, 3 behind its 4 own rows
            w1 = torch.arange((n_own + 2) * width, dtype=torch.float32).view(-1, width) + 2
            want[1] += w1[n_own]
            want[3] += w1[n_own + 1]
        if rank == 1:  # rank 0 holds a copy of row 2
            w0 = torch.arange((n_own + 1) * width, dtype=torch.float32).view(-1, width) + 1
            want[2] += w0[n_own]
        torch.save(dict(fwd=fwd_calls, bwd=len(calls) - fwd_calls, err=float((rows.grad - want).abs().max()),
                        n_full=full.shape[0]), f"{result_file}.{rank}")
    finally:
        dist.destroy_process_group()


def test_halo_backward_is_entered_by_a_rank_with_an_empty_halo(tmp_path):
    """The reverse all-to-all-v of ``_HaloRows.backward`` is a group-wide collective: a rank that neither sends nor
    receives boundary rows must still enter it (zero-length splits), as it does in the forward -- on RCCL a skipped call
    hangs or mis-pairs the peers' collective.  World 3, last rank with an empty halo: one exchange per direction on EVERY
    rank, gradients = own rows + the peers' halo copies."""
    port = 29500 + (os.getpid() % 200)
    result = str(tmp_path / "res")
    mp.spawn(_empty_halo_worker, args=(3, port, result), nprocs=3, join=True)
    for r in range(3):
        info = torch.load(f"{result}.{r}")
        assert info["fwd"] == 1 and info["bwd"] == 1, (r, info)
        assert info["err"] == 0.0, (r, info)
    assert torch.load(f"{result}.2")["n_full"] == 4
