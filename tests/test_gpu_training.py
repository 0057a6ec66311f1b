"""Training of every model family that has a backward on the HIP kernels: whole-model output and
every parameter gradient against torch autograd through the CPU oracle (f64), on the golden weights of the reference.

Flat GraphTransformer model: tests/test_gpu_parity.py (``test_whole_model_training_step_*``).  Here: GNN processor with
GraphTransformer mappers, GNN processor + GNN mappers, the hierarchical model, block-level ``.backward()`` as the
reference's own tests do it, activation checkpointing on / off, boundings under autograd.
"""

import pytest
import torch

from conftest import same_bits_or_last_bit_rows
from conftest import split_prefix
from oracle import reference_path as ref
from test_oracle_golden import graph_tensors
from test_oracle_golden import hier_graph_tensors

pytestmark = pytest.mark.gpu

DEV = "cuda"
KW = dict(num_heads=16, num_layers=4, num_chunks=2, prognostic_in=list(range(10)), prognostic_out=list(range(10)))


def rel_err(got, want):
    got, want = got.float().cpu(), want.float().cpu()
    return float((got - want).abs().max() / want.abs().max().clamp_min(1e-30))


def _oracle_grads(fn, sd, x, dy):
    rsd = {k: (v.double().requires_grad_() if v.is_floating_point() else v) for k, v in sd.items()}
    y = fn(rsd, x.double())
    y.backward(dy.double())
    return y.detach().float(), rsd


def _compare_grads(model, rsd, tol=5e-3):
    grads = dict(model.named_parameters())
    used = [k for k in grads if rsd[k].grad is not None and float(rsd[k].grad.abs().max()) > 0]
    scale_all = max(float(rsd[k].grad.abs().max()) for k in used)
    for k in used:
        assert grads[k].grad is not None, k
        err = float((grads[k].grad.cpu() - rsd[k].grad.float()).abs().max())
        assert err <= tol * max(float(rsd[k].grad.abs().max()), 0.02 * scale_all), (k, err, float(rsd[k].grad.abs().max()))
    return used


def _compare_block_grads(blk, rsd, tol=5e-3):
    scale_all = max(float(v.grad.abs().max()) for v in rsd.values() if v.grad is not None)
    for k, p in blk.named_parameters():
        want = rsd["x." + k].grad
        assert p.grad is not None and want is not None, k
        err = float((p.grad.cpu() - want.float()).abs().max())
        assert err <= tol * max(float(want.abs().max()), 0.02 * scale_all), (k, err)


def _f64(graph):
    return {k: (v.double() if v.is_floating_point() else v) for k, v in graph.items()}


@pytest.mark.parametrize("mappers", ["GraphTransformer", "GNN"])
def test_gnn_model_training_step_vs_oracle_autograd(graph_o32, golden_cfg1_gnn, golden_cfg1_gnn_all, mappers):
    """GNN processor (edge-MLP message passing) with GraphTransformer or GNN mappers: forward + backward through the
    nn.Module on the HIP kernels (gather_add_act / segment_sum and their backward, fused Linear, LayerNorm)."""
    from test_gpu_parity import _build

    gold = golden_cfg1_gnn if mappers == "GraphTransformer" else golden_cfg1_gnn_all
    sd = split_prefix(gold, "sd.")
    graph = _f64(graph_tensors(graph_o32))
    dy = torch.randn(gold["y"].shape, generator=torch.Generator().manual_seed(2))
    want, rsd = _oracle_grads(lambda s, xx: ref.model_forward(s, graph, xx, processor="GNN", mappers=mappers, **KW), sd,
                              gold["x"], dy)
    assert rel_err(want, gold["y"]) < 1e-4
    model, _ = _build(graph_o32, 64, 4, processor="GNN", mappers=mappers)
    model.load_state_dict(sd)
    model = model.to(DEV)
    y = model(gold["x"].to(DEV))
    assert y.requires_grad and rel_err(y.detach(), gold["y"]) < 1e-4
    y.backward(dy.to(DEV))
    used = _compare_grads(model, rsd)
    assert len(used) > 60 and any("conv.edge_mlp" in k for k in used) and any(k.endswith("trainable") for k in used)
    torch.optim.SGD(model.parameters(), lr=1e-3).step()


def test_hierarchical_model_training_step_vs_oracle_autograd(graph_hier, golden_hier_gt):
    from test_gpu_parity import _build_hier

    gold = golden_hier_gt
    sd = split_prefix(gold, "sd.")
    graph = _f64(hier_graph_tensors(graph_hier))
    dy = torch.randn(gold["y"].shape, generator=torch.Generator().manual_seed(3))
    want, rsd = _oracle_grads(
        lambda s, xx: ref.hierarchical_forward(s, graph, xx, hidden=["hidden_1", "hidden_2"], num_heads=16, level_layers=2,
                                               prognostic_in=list(range(10)), prognostic_out=list(range(10))),
        sd, gold["x"], dy)
    model = _build_hier(graph_hier)
    model.load_state_dict(sd)
    model = model.to(DEV)
    y = model(gold["x"].to(DEV))
    assert y.requires_grad and rel_err(y.detach(), gold["y"]) < 1e-4
    y.backward(dy.to(DEV))
    used = _compare_grads(model, rsd)
    assert any(k.startswith("downscale.") for k in used) and any(k.startswith("up_level_processor.") for k in used)


def test_block_level_backward_like_the_reference_tests(golden_blocks):
    """reference tests/layers/processor/test_graphtransformer_processor.py:145-159 style: call the block, ``.backward()``
    on a sum, every parameter has a gradient -- GraphTransformer processor block, mapper block and GNN block, checked
    against the oracle's autograd."""
    from anemoi_models_amd.layers.block import GraphConvProcessorBlock
    from anemoi_models_amd.layers.block import GraphTransformerMapperBlock
    from anemoi_models_amd.layers.block import GraphTransformerProcessorBlock

    b = golden_blocks
    # --- GraphTransformer processor block (128 channels, 16 heads of 8: f32 lanes own 4 channels)
    sd = split_prefix(b, "gtp.sd.")
    blk = GraphTransformerProcessorBlock(128, 512, 128, edge_dim=b["gtp.edge_attr"].shape[1], num_heads=16)
    blk.load_state_dict(sd)
    blk = blk.to(DEV)
    x = b["gtp.x"].to(DEV).requires_grad_()
    y, ea = blk(x, b["gtp.edge_attr"].to(DEV), b["gtp.edge_index"].to(DEV), None, 1)
    assert rel_err(y.detach(), b["gtp.y"]) < 1e-4
    y.sum().backward()
    rsd = {"x." + k: v.double().requires_grad_() for k, v in sd.items()}
    xr = b["gtp.x"].double().requires_grad_()
    ref.gt_processor_block(rsd, "x", xr, b["gtp.edge_attr"].double(), b["gtp.edge_index"], 16).sum().backward()
    assert rel_err(x.grad, xr.grad) < 2e-3
    _compare_block_grads(blk, rsd)  # (lin_key.bias has an analytically ZERO gradient: scaled, not relative, comparison)
    # --- GraphTransformer mapper block
    sd = split_prefix(b, "gtm.sd.")
    blk = GraphTransformerMapperBlock(64, 256, 64, edge_dim=b["gtm.edge_attr"].shape[1], num_heads=16)
    blk.load_state_dict(sd)
    blk = blk.to(DEV)
    xs, xd = b["gtm.x_src"].to(DEV).requires_grad_(), b["gtm.x_dst"].to(DEV).requires_grad_()
    (_, y), _ = blk((xs, xd), b["gtm.edge_attr"].to(DEV), b["gtm.edge_index"].to(DEV), None, 1,
                    size=(xs.shape[0], xd.shape[0]))
    assert rel_err(y.detach(), b["gtm.y_dst"]) < 1e-4
    y.sum().backward()
    assert xs.grad is not None and xd.grad is not None and all(p.grad is not None for p in blk.parameters())
    # --- GNN processor block: edges in the caller's order in and out
    sd = split_prefix(b, "gnn.sd.")
    blk = GraphConvProcessorBlock(64, 64, mlp_extra_layers=0, activation="SiLU")
    blk.load_state_dict(sd)
    blk = blk.to(DEV)
    x, e = b["gnn.x"].to(DEV).requires_grad_(), b["gnn.edge_attr"].to(DEV).requires_grad_()
    xn, en = blk(x, e, b["gnn.edge_index"].to(DEV), None)
    assert rel_err(xn.detach(), b["gnn.y"]) < 1e-4 and rel_err(en.detach(), b["gnn.edges_new"]) < 1e-4
    (xn.sum() + en.sum()).backward()
    rsd = {"x." + k: v.double().requires_grad_() for k, v in sd.items()}
    xr, er = b["gnn.x"].double().requires_grad_(), b["gnn.edge_attr"].double().requires_grad_()
    xo, eo = ref.gnn_processor_block(rsd, "x", xr, er, b["gnn.edge_index"])
    (xo.sum() + eo.sum()).backward()
    assert rel_err(x.grad, xr.grad) < 2e-3 and rel_err(e.grad, er.grad) < 2e-3
    _compare_block_grads(blk, rsd)


def test_mapper_block_update_src_nodes_trains(golden_blocks):
    """``GraphTransformerMapperBlock(update_src_nodes=True)`` (reference layers/block.py:540-546: ``node_src_mlp(x_src) +
    x_src``, row-local): the differentiable route returns the updated sources -- same values as the inference route, the
    gradients of the source MLP against torch's own autograd of that stack in f64."""
    from anemoi_models_amd.layers.block import GraphTransformerMapperBlock

    b = golden_blocks
    torch.manual_seed(11)
    blk = GraphTransformerMapperBlock(64, 256, 64, edge_dim=b["gtm.edge_attr"].shape[1], num_heads=16,
                                      update_src_nodes=True).to(DEV)
    ea, ei = b["gtm.edge_attr"].to(DEV), b["gtm.edge_index"].to(DEV)
    xs, xd = b["gtm.x_src"].to(DEV), b["gtm.x_dst"].to(DEV)
    size = (xs.shape[0], xd.shape[0])
    with torch.no_grad():
        (want_s, want_d), _ = blk.eval()((xs, xd), ea, ei, None, 1, size=size)
    blk.train()
    xs_g, xd_g = xs.clone().requires_grad_(), xd.clone().requires_grad_()
    (got_s, got_d), _ = blk((xs_g, xd_g), ea, ei, None, 1, size=size)
    assert rel_err(got_s.detach(), want_s) < 1e-5 and rel_err(got_d.detach(), want_d) < 1e-5
    assert rel_err(got_s.detach(), xs) > 1e-3  # (really updated)
    w = torch.randn_like(got_s)
    ((got_s * w).sum() + got_d.sum()).backward()
    import copy

    mlp64 = copy.deepcopy(blk.node_src_mlp).double()
    for q in mlp64.parameters():
        q.grad = None
    xr = xs.double().requires_grad_()
    ((mlp64(xr) + xr) * w.double()).sum().backward()
    for (name, p_), q in zip(blk.node_src_mlp.named_parameters(), mlp64.parameters()):
        assert p_.grad is not None and rel_err(p_.grad, q.grad) < 2e-3, name
    # d x_src = the source MLP's path + the k / v path of the attention: the former alone when the destination loss is dropped
    blk.zero_grad()
    xs_g2 = xs.clone().requires_grad_()
    (got_s2, _), _ = blk((xs_g2, xd), ea, ei, None, 1, size=size)
    (got_s2 * w).sum().backward()
    assert rel_err(xs_g2.grad, xr.grad) < 2e-3


def test_checkpointing_reproduces_the_gradients_bit_for_bit(graph_o32, golden_cfg1_gt, monkeypatch):
    """Mapper calls and processor chunks are recomputed in the backward (as the reference checkpoints them); all kernels
    are deterministic, so the gradients equal those of the run that kept every activation -- and less memory is held."""
    from test_gpu_parity import _build

    gold = golden_cfg1_gt
    dy = torch.randn(gold["y"].shape, generator=torch.Generator().manual_seed(2)).to(DEV)
    res = {}
    # "1": every region but the last (the decoder keeps its activations: recomputing it at the start of the backward could
    # not lower the peak); "all": the decoder wrapped too, as the reference does; "0": nothing checkpointed
    for mode in ("1", "all", "0"):
        monkeypatch.setenv("ANEMOI_AMD_CHECKPOINT", "0" if mode == "0" else "1")
        monkeypatch.setenv("ANEMOI_AMD_CHECKPOINT_LAST", "1" if mode == "all" else "0")
        model, _ = _build(graph_o32, 64, 4)
        model.load_state_dict(split_prefix(gold, "sd."))
        model = model.to(DEV)
        base = torch.cuda.memory_allocated()
        y = model(gold["x"].to(DEV))
        held = torch.cuda.memory_allocated() - base
        torch.cuda.reset_peak_memory_stats()
        y.backward(dy)
        peak = torch.cuda.max_memory_allocated() - base
        res[mode] = (y.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters()}, held, peak)
        del model, y
    for mode in ("1", "all"):
        assert torch.equal(res[mode][0], res["0"][0])
        for k, g in res[mode][1].items():
            assert torch.equal(g, res["0"][1][k]), (mode, k)
    assert res["all"][2] < 0.6 * res["0"][2], (res["all"][2], res["0"][2])  # held between forward and backward
    assert res["1"][2] < res["0"][2]
    print("peak bytes in the backward: decoder kept", res["1"][3], "every region recomputed", res["all"][3], "nothing", res["0"][3])
    assert res["1"][3] <= 1.02 * res["all"][3], (res["1"][3], res["all"][3])  # keeping the last region costs nothing at the peak


@pytest.mark.parametrize("checkpoint", ["1", "0"])
@pytest.mark.parametrize("channels,heads", [(128, 16), (64, 16)])
def test_batched_processor_weights_equal_the_per_block_route(graph_o32, monkeypatch, checkpoint, channels, heads):
    """bf16 training of a GraphTransformer processor prepares what its blocks derive from their parameters (lin_edge fold,
    weight assembly, casts, transposes) for ALL blocks at once (``autograd.gt_processor_weights``) and reduces the weight
    gradients straight into a stacked buffer (``autograd.GradSink``).  Same kernels on the same operands as the per-block
    route (``ANEMOI_AMD_TRAIN_BATCHED_PARAMS=0``): prediction and gradients agree to the rounding of the batched einsums."""
    from test_gpu_parity import _build
    from anemoi_models_amd import autograd

    monkeypatch.setenv("ANEMOI_AMD_DTYPE", "bf16")
    monkeypatch.setenv("ANEMOI_AMD_CHECKPOINT", checkpoint)
    torch.manual_seed(11)
    model, _ = _build(graph_o32, channels, 4, heads=heads)
    model = model.to(DEV).train()
    x = torch.randn((1, 2, 1, graph_o32["data"].num_nodes, 12), generator=torch.Generator().manual_seed(3)).to(DEV)
    calls = []
    real = autograd.gt_processor_weights
    monkeypatch.setattr(autograd, "gt_processor_weights", lambda *a, **k: calls.append(real(*a, **k)) or calls[-1])
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("ANEMOI_AMD_TRAIN_BATCHED_PARAMS", mode)
        for p in model.parameters():
            p.grad = None
        y = model(x)
        dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(2)).to(DEV)
        y.backward(dy)
        res[mode] = (y.detach().float().clone(), {k: p.grad.float().clone() for k, p in model.named_parameters()
                                                   if p.grad is not None})
    assert calls[0] is not None and len(calls[0]) == 4 and calls[1] is None  # the batched route ran / was switched off
    y_err = rel_err(res["1"][0], res["0"][0])
    assert set(res["1"][1]) == set(res["0"][1])
    scale_all = max(float(g.abs().max()) for g in res["0"][1].values())
    worst = 0.0
    for k, g0 in res["0"][1].items():
        err = float((res["1"][1][k] - g0).abs().max())
        worst = max(worst, err / max(float(g0.abs().max()), 0.05 * scale_all))
    print(f"batched vs per-block parameters: prediction rel err {y_err:.2e}, worst gradient rel err {worst:.2e}")
    # (the folded u / t weights come out of a batched einsum here: a last-bit difference in f32 may round to the other
    #  bf16 neighbour; everything else is the same kernel on the same operands)
    assert y_err <= 2e-3 and worst <= 2e-2, (y_err, worst)


def test_wide_dx_and_fused_finish_equal_the_plain_routes(graph_o32, golden_cfg1_gt, monkeypatch):
    """Two training-step shortcuts against the routes they replace: (a) the dX GEMM of a Linear with few input features on very
    many rows runs on the persistent kernel through a transposed weight padded to 256 zero rows (``ANEMOI_AMD_TRAIN_WIDE_DX``);
    (b) the prognostic residual of the model output is ONE pass of ``anemoi_finalize_output`` under autograd
    (``ANEMOI_AMD_TRAIN_FUSED_FINISH``)."""
    from test_gpu_parity import _build
    from anemoi_models_amd import autograd

    g = torch.Generator().manual_seed(4)
    m, k_in, n = 70001, 192, 1024
    x = torch.randn(m, k_in, generator=g).bfloat16().to(DEV)
    w = (torch.randn(n, k_in, generator=g) / k_in**0.5).to(DEV)
    b = torch.randn(n, generator=g).to(DEV)
    dy = torch.randn(m, n, generator=g).bfloat16().to(DEV)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("ANEMOI_AMD_TRAIN_WIDE_DX", mode)
        xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
        y = autograd.linear(xr, wr, br)
        y.backward(dy)
        res[mode] = (y.detach().clone(), xr.grad.clone(), wr.grad.clone(), br.grad.clone())
    assert torch.equal(res["1"][0], res["0"][0]) and torch.equal(res["1"][2], res["0"][2]) and torch.equal(res["1"][3], res["0"][3])
    want = dy.float() @ w
    assert res["1"][1].shape == (m, k_in) and rel_err(res["1"][1], want) < 1e-2 and rel_err(res["0"][1], want) < 1e-2
    assert rel_err(res["1"][1], res["0"][1]) < 1e-2

    gold = golden_cfg1_gt
    dyo = torch.randn(gold["y"].shape, generator=torch.Generator().manual_seed(2)).to(DEV)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("ANEMOI_AMD_TRAIN_FUSED_FINISH", mode)
        model, _ = _build(graph_o32, 64, 4)
        model.load_state_dict(split_prefix(gold, "sd."))
        model = model.to(DEV).train()
        y = model(gold["x"].to(DEV))
        y.backward(dyo)
        out[mode] = (y.detach().clone(), {k_: p.grad.clone() for k_, p in model.named_parameters() if p.grad is not None})
    assert out["1"][0].dtype == out["0"][0].dtype == torch.float32 and rel_err(out["1"][0], gold["y"]) < 1e-4
    assert torch.allclose(out["1"][0], out["0"][0], rtol=0, atol=1e-6)
    assert set(out["1"][1]) == set(out["0"][1])
    for k_, g1 in out["1"][1].items():
        assert torch.equal(g1, out["0"][1][k_]), k_

    # (c) bf16: the model input written once in the layout of the mappers' first GEMMs ([x | latlon | trainable | 1 | 0-pad],
    # training._AssembleNodes, ANEMOI_AMD_TRAIN_ASSEMBLE) against permute / cat / cast + one concatenation per folded embedding:
    # with the fold (128 channels) and without it (64 channels: the embeddings stay GEMMs on the wider rows)
    monkeypatch.setenv("ANEMOI_AMD_DTYPE", "bf16")
    for channels in (128, 64):
        torch.manual_seed(12)
        model, _ = _build(graph_o32, channels, 2)
        model = model.to(DEV).train()
        xin = gold["x"].to(DEV)
        res_c = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("ANEMOI_AMD_TRAIN_ASSEMBLE", mode)
            for p_ in model.parameters():
                p_.grad = None
            y = model(xin)
            y.backward(dyo)
            res_c[mode] = (y.detach().clone(), {k_: p_.grad.float().clone() for k_, p_ in model.named_parameters()
                                                 if p_.grad is not None})
        assert rel_err(res_c["1"][0], res_c["0"][0]) < 2e-3, channels
        assert set(res_c["1"][1]) == set(res_c["0"][1])
        scale_all = max(float(g0.abs().max()) for g0 in res_c["0"][1].values())
        for k_, g0 in res_c["0"][1].items():
            err = float((res_c["1"][1][k_] - g0).abs().max())
            assert err <= 2e-2 * max(float(g0.abs().max()), 0.05 * scale_all), (channels, k_, err, float(g0.abs().max()))


def test_bf16_training_at_head_size_4(graph_o32, golden_cfg1_gt, monkeypatch):
    """BASELINE config 1 (64 channels, 16 heads: head size 4) trains in bf16: the bf16 edge kernels move 8 channels per lane,
    so the edge phases of its blocks run on the f32 kernels between two casts (``autograd._edge_phase_in_f32``), every GEMM
    stays bf16.  Loss and gradients against the f32 training step of the same weights (itself held to the oracle's autograd
    by the tests above); blocks with the explicit-edge conv (edge_dim 39) against the oracle directly."""
    from test_gpu_parity import _build, split_prefix
    from anemoi_models_amd.layers.block import GraphTransformerProcessorBlock

    gold = golden_cfg1_gt
    x = gold["x"].to(DEV)
    dy = torch.randn(gold["y"].shape, generator=torch.Generator().manual_seed(2)).to(DEV)
    res = {}
    for mode in ("fp32", "bf16"):
        monkeypatch.setenv("ANEMOI_AMD_DTYPE", mode)
        model, _ = _build(graph_o32, 64, 4)
        model.load_state_dict(split_prefix(gold, "sd."))
        model = model.to(DEV).train()
        y = model(x)
        y.backward(dy)
        res[mode] = (y.detach().float().clone(), {k: p.grad.float().clone() for k, p in model.named_parameters()
                                                   if p.grad is not None})
    assert rel_err(res["fp32"][0], gold["y"]) < 1e-4
    assert rel_err(res["bf16"][0], gold["y"]) < 5e-2
    assert set(res["bf16"][1]) == set(res["fp32"][1])
    scale_all = max(float(g.abs().max()) for g in res["fp32"][1].values())
    for k, g32 in res["fp32"][1].items():
        err = float((res["bf16"][1][k] - g32).abs().max())
        assert err <= 8e-2 * max(float(g32.abs().max()), 0.05 * scale_all), (k, err, float(g32.abs().max()))
    # explicit per-edge features (edge_dim beyond the fold), head size 4, bf16
    g = torch.Generator().manual_seed(40)
    c, h, edge_dim, n, e = 64, 16, 39, 120, 900
    torch.manual_seed(6)
    blk = GraphTransformerProcessorBlock(c, 2 * c, c, edge_dim=edge_dim, num_heads=h)
    ei = torch.stack([torch.randint(0, n, (e,), generator=g), torch.randint(0, n - 1, (e,), generator=g)])
    x0, ea0 = torch.randn(n, c, generator=g), torch.randn(e, edge_dim, generator=g)
    rsd = {"x." + k: v.detach().double().requires_grad_() for k, v in blk.named_parameters()}
    xr = x0.double().requires_grad_()
    want = ref.gt_processor_block(rsd, "x", xr, ea0.double(), ei, h)
    want.sum().backward()
    blk = blk.to(DEV)
    xg = x0.to(DEV).requires_grad_()
    yb, _ = blk(xg, ea0.to(DEV), ei.to(DEV), None, 1)
    assert rel_err(yb.detach(), want.detach()) < 8e-2
    yb.sum().backward()
    assert rel_err(xg.grad, xr.grad) < 8e-2
    _compare_block_grads(blk, rsd, tol=8e-2)


def test_cpu_offload_keeps_the_gradients(graph_o32, golden_cfg1_gt):
    """``cpu_offload=True`` of the mappers / processor (reference layers/mapper.py:64-66, layers/processor.py:65-67:
    ``offload_wrapper``): the tensors saved for the backward travel through pinned host memory; loss and every parameter
    gradient are bit-identical to the run without it."""
    from test_gpu_parity import _build, split_prefix

    model, _ = _build(graph_o32, 64, 4)
    model.load_state_dict(split_prefix(golden_cfg1_gt, "sd."))
    model = model.to(DEV).train()
    x = golden_cfg1_gt["x"].to(DEV)
    params = [p for p in model.parameters() if p.requires_grad]

    def step():
        for p in params:
            p.grad = None
        loss = (model(x) ** 2).mean()
        loss.backward()
        return loss.detach().clone(), [p.grad.clone() for p in params]

    base_loss, base_grads = step()
    for m in (model.encoder, model.processor, model.decoder):
        m.offload_layers(True)
    off_loss, off_grads = step()
    assert torch.equal(off_loss, base_loss)
    for a, b in zip(off_grads, base_grads):
        assert torch.equal(a, b)


@pytest.mark.parametrize("checkpoint", ["1", "0"])
def test_graphed_train_step_with_attention_dropout_equals_eager(graph_o32, monkeypatch, checkpoint):
    """``TransformerProcessor``'s default attention dropout (reference layers/processor.py:99: 0.1) inside a captured step:
    the per-step part of the mask seed is a counter in DEVICE memory the graph itself advances (runtime.DeviceDropout), so
    replays draw new masks -- and an eager loop in the same context, started at the same counter value, draws the same
    ones: loss and every parameter gradient bit for bit, over three steps, with and without activation checkpointing (the
    recomputed forward must rebuild the step's mask, not the next one)."""
    from test_gpu_parity import _build

    from anemoi_models_amd.runtime import DeviceDropout, GraphedTrainStep

    monkeypatch.setenv("ANEMOI_AMD_CHECKPOINT", checkpoint)
    torch.manual_seed(5)
    model, idx = _build(graph_o32, 64, 2, processor="Transformer")
    model = model.to(DEV).train()
    for m in model.modules():
        if hasattr(m, "dropout_p"):
            m.dropout_p = 0.1
    n = graph_o32["data"].num_nodes
    g = torch.Generator().manual_seed(9)
    xs = [torch.randn(1, 2, 1, n, idx.num_input, generator=g).to(DEV) for _ in range(3)]
    ts = [torch.randn(1, 1, n, model.num_output_channels, generator=g).to(DEV) for _ in range(3)]
    loss_fn = lambda y, tt: ((y - tt) ** 2).mean()  # noqa: E731
    step = GraphedTrainStep(model, loss_fn, xs[0], ts[0])
    assert step.dropout is not None
    params = [p for p in model.parameters() if p.requires_grad]
    step.dropout.counter.fill_(100)
    graphed = []
    for x, t in zip(xs, ts):
        loss = step(x, t)
        graphed.append((loss.clone(), [p.grad.clone() for p in params]))
    assert int(step.dropout.counter.item()) == 103
    assert not torch.equal(graphed[0][0], step(xs[0], ts[0]))  # the same batch again: another mask
    # the eager loop: same modules (same per-layer seed constants), same counter start
    with DeviceDropout(DEV, start=100) as dd:
        for (x, t), (g_loss, g_grads) in zip(zip(xs, ts), graphed):
            for p in params:
                p.grad = None
            dd.advance()
            loss = loss_fn(model(x), t)
            loss.backward()
            assert torch.equal(loss.detach(), g_loss)
            for p, gg in zip(params, g_grads):
                assert torch.equal(p.grad, gg)
    # the frozen configuration is checked at every call
    model.eval()
    with pytest.raises(RuntimeError, match="changed after the capture"):
        step(xs[0], ts[0])


@pytest.mark.parametrize("checkpoint", ["1", "0"])
def test_training_step_captured_in_a_hip_graph(graph_o32, golden_cfg1_gt, monkeypatch, checkpoint):
    """runtime.GraphedTrainStep: forward + loss + backward captured once, replayed on new inputs -- loss and every
    parameter gradient equal the eager step's bit for bit (the kernels are deterministic and the graph replays the same
    launches); with a capturable optimizer inside the graph, two replays equal two eager steps."""
    from test_gpu_parity import _build

    from anemoi_models_amd.runtime import GraphedTrainStep

    monkeypatch.setenv("ANEMOI_AMD_CHECKPOINT", checkpoint)
    gold = golden_cfg1_gt
    gen = torch.Generator().manual_seed(5)
    xs = [gold["x"].to(DEV), (gold["x"] + 0.1 * torch.randn(gold["x"].shape, generator=gen)).to(DEV)]
    ts = [torch.randn(gold["y"].shape, generator=gen).to(DEV) for _ in xs]
    loss_fn = lambda y, t: ((y - t) ** 2).mean()  # noqa: E731

    def fresh():
        model, _ = _build(graph_o32, 64, 4)
        model.load_state_dict(split_prefix(gold, "sd."))
        return model.to(DEV).train()

    # gradients only.  (The eager steps live in a function: an autograd graph of an earlier step that is still referenced --
    # a kept ``loss`` -- pins the parameters' AccumulateGrad nodes to the stream it was built on, and a capture that
    # reuses them touches that stream: torch's whole-network capture rule, see GraphedTrainStep.)
    model = fresh()

    def eager_step(x, t):
        for p in model.parameters():
            p.grad = None
        loss = loss_fn(model(x), t)
        loss.backward()
        return loss.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}

    want = [eager_step(x, t) for x, t in zip(xs, ts)]
    graphed = GraphedTrainStep(model, loss_fn, torch.zeros_like(xs[0]), torch.zeros_like(ts[0]))
    for (x, t), (wl, wg) in zip(zip(xs, ts), want):
        gl = graphed(x, t)
        assert torch.equal(gl, wl)
        got = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
        assert set(got) == set(wg) and len(wg) > 20
        for k, g in wg.items():
            assert torch.equal(got[k], g), k

    # optimizer step inside the graph
    eager, captured = fresh(), fresh()
    opt_e = torch.optim.SGD(eager.parameters(), lr=1e-2, momentum=0.9)
    opt_c = torch.optim.SGD(captured.parameters(), lr=1e-2, momentum=0.9)
    graphed = GraphedTrainStep(captured, loss_fn, xs[0], ts[0], optimizer=opt_c, warmup=1)
    # the warm-up step is a real optimizer step on the example; bring the eager model to the same state
    opt_e.zero_grad(set_to_none=True)  # (the capture pass records, it does not run: one step, not two)
    loss_fn(eager(xs[0]), ts[0]).backward()
    opt_e.step()

    def eager_opt_step(x, t):
        opt_e.zero_grad(set_to_none=True)
        le = loss_fn(eager(x), t)
        le.backward()
        opt_e.step()
        return le.detach()

    for x, t in zip(xs, ts):
        le = eager_opt_step(x, t)
        lc = graphed(x, t)
        assert rel_err(lc.detach(), le) < 1e-6
    for (k, pe), (_, pc) in zip(eager.named_parameters(), captured.named_parameters()):
        assert rel_err(pc.detach(), pe.detach()) < 1e-5, k


def test_training_with_boundings(graph_o32, golden_cfg1_gt):
    """A model with a ``bounding:`` list under autograd: the forward equals the reference golden output and the clamps
    cut the gradient where they bite."""
    from conftest import load_npz
    from test_bounding import bounded_model

    model = bounded_model(graph_o32)
    model.load_state_dict(split_prefix(golden_cfg1_gt, "sd."))
    model = model.to(DEV)
    y = model(golden_cfg1_gt["x"].to(DEV))
    assert y.requires_grad and rel_err(y.detach(), load_npz("bounding_gt.npz")["y"]) < 1e-4
    y.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters() if p.requires_grad)


@pytest.mark.parametrize("world,graph_name,channels,layers,heads,family", [
    (2, "o32_ico2", 64, 4, 4, "GraphTransformer"), (3, "o48_ico3", 128, 2, 8, "GraphTransformer"),
    (2, "o32_ico2", 64, 2, 4, "GNN"),       # GNN processor between GraphTransformer mappers
    (3, "o32_ico2", 64, 2, 4, "GNN_all"),   # GNN processor and GNN mappers
    (2, "o32_ico2", 64, 2, 4, "Transformer"),  # rows <-> heads exchanges around the attention, as autograd nodes
])
def test_node_partitioned_training_step_ranks_sharing_one_gpu(world, graph_name, channels, layers, heads, family, tmp_path):
    """Training across a model group: the sharded differentiable forward (halo all-to-all-v per block, output all-gather)
    and its backward (reverse halo all-to-all-v + index-add, gradient slice of the gather) on the HIP kernels, the ranks
    as separate processes sharing cuda:0 with host-staged gloo collectives.  Output == single-device output; parameter
    gradients summed over the ranks == single-device gradients (reference distributed/graph.py:152-162 semantics)."""
    import os
    import subprocess
    import sys

    port = 29500 + (os.getpid() % 150)
    out = str(tmp_path / "res")
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_gpu_shared_ranks.py")
    env = dict(os.environ, ANEMOI_TEST_FAMILY=family)
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), str(port), out, graph_name, str(channels),
                               str(layers), str(heads), "fp32", "train"], env=env) for r in range(world)]
    try:
        codes = [p.wait(timeout=900) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
    assert codes == [0] * world
    for r in range(world):
        i = torch.load(f"{out}.{r}")
        assert i["requires_grad"] and i["n_grads"] > 40
        assert i["err"] <= 2e-5 * max(1.0, i["scale"]), i
        assert i["train_out_err"] <= 2e-5 * max(1.0, i["scale"]), i
        assert i["grad_err"] <= 2e-4 * i["grad_scale"], i


@pytest.mark.parametrize("dtype,b,s,h,d,window", [
    (torch.float32, 2, 300, 4, 64, -1), (torch.float32, 1, 257, 2, 24, 30), (torch.bfloat16, 1, 700, 8, 64, -1),
    (torch.bfloat16, 2, 333, 16, 32, -1), (torch.bfloat16, 1, 400, 2, 64, 50), (torch.bfloat16, 1, 1026, 2, 64, -1),
    # windows that cut the tile loops of both backward kernels at both ends (LDS ring started at a tile > 0)
    (torch.bfloat16, 1, 3000, 2, 64, 100), (torch.bfloat16, 2, 1500, 4, 32, 70),
])
def test_mhsa_backward_vs_torch_autograd(dtype, b, s, h, d, window):
    """anemoi_mhsa_backward (probabilities recomputed from the forward's log-sum-exp) against torch autograd through an
    f64 softmax(QK^T / sqrt(D)) V -- MFMA forward (bf16, D = 64 / 32) and the generic forward, global and windowed."""
    from anemoi_models_amd import autograd

    g = torch.Generator().manual_seed(s + d)
    c = h * d
    qkv = (torch.randn(b * s, 3 * c, generator=g) * 0.8).to(dtype)
    dout = torch.randn(b * s, c, generator=g).to(dtype)
    ref_in = qkv.double().requires_grad_()
    q, k, v = (t.reshape(b, s, h, d).permute(0, 2, 1, 3) for t in ref_in.split(c, dim=1))
    sc = q @ k.transpose(-1, -2) / d**0.5
    if window >= 0:
        i = torch.arange(s)
        sc = sc.masked_fill((i[:, None] - i[None, :]).abs() > window, float("-inf"))
    want = (torch.softmax(sc, -1) @ v).permute(0, 2, 1, 3).reshape(b * s, c)
    want.backward(dout.double())
    x = qkv.to(DEV).requires_grad_()
    got = autograd.mhsa(x, b, h, window)
    got.backward(dout.to(DEV))
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert rel_err(got.detach(), want.detach()) < tol
    assert rel_err(x.grad, ref_in.grad) < (1e-4 if dtype == torch.float32 else 3e-2)


@pytest.mark.parametrize("d", [64, 32])
@pytest.mark.parametrize("b,s", [(1, 257), (1, 512), (2, 1301), (1, 2112)])   # ragged / tile-aligned / batched / 33 tiles
@pytest.mark.parametrize("window", [-1, 100])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_mhsa_backward_mfma_route_vs_valu_route(d, b, s, window, p):
    """The MFMA backward (inline-asm products with hand-counted wait states the compiler cannot check) against the VALU
    backward of the same entry point (plain HIP: every hazard is the compiler's) -- same inputs, same forward statistics,
    same dropout mask: a compiler upgrade or a schedule change that breaks a wait state shows here as a wrong gradient,
    on a grid of head size x ragged / aligned / batched S x window x dropout."""
    from anemoi_models_amd import _lib, ops

    h = 4
    c = h * d
    g = torch.Generator().manual_seed(1000 * d + s + window + int(100 * p))
    qkv = (torch.randn(b * s, 3 * c, generator=g) * 0.9).bfloat16().to(DEV)
    dout = torch.randn(b * s, c, generator=g).bfloat16().to(DEV)
    out, lse = ops.mhsa(qkv, b, h, window, return_lse=True, dropout_p=p, dropout_seed=77)
    fast = ops.mhsa_backward(qkv, out, dout, lse, b, h, window, dropout_p=p, dropout_seed=77)
    slow = ops.mhsa_backward(qkv, out, dout, lse, b, h, window, dropout_p=p, dropout_seed=77, use_mfma=False)
    assert torch.isfinite(fast.float()).all() and torch.isfinite(slow.float()).all()
    for i, name in enumerate(("dq", "dk", "dv")):
        a, w = fast[:, i * c:(i + 1) * c], slow[:, i * c:(i + 1) * c]
        assert rel_err(a, w) < 2e-2, (name, rel_err(a, w))
        rows = (a.float() - w.float()).abs().amax(1) / w.float().abs().amax(1).clamp_min(1e-3 * float(w.float().abs().max()))
        assert float(rows.max()) < 6e-2, (name, int(rows.argmax()), float(rows.max()))
    assert b"clang" in _lib.load().anemoi_build_info() or b"unknown" in _lib.load().anemoi_build_info()


@pytest.mark.parametrize("s,h,d,window,p", [(2100, 4, 64, -1, 0.0), (1300, 8, 32, -1, 0.0), (1700, 2, 64, 90, 0.1)])
def test_mhsa_backward_repeated_calls_are_bit_identical(s, h, d, window, p):
    """The backward kernels stream their tiles through an LDS-DMA ring (round 3): 40 calls on the same inputs, with the
    allocator churning in between, must give bit-identical gradients (a tile read before it has landed shows up as a
    difference between calls long before it shows up against the reference)."""
    import random

    from anemoi_models_amd import ops

    random.seed(s)
    c = h * d
    g = torch.Generator().manual_seed(s + d)
    qkv_cpu = (torch.randn(s, 3 * c, generator=g) * 0.8).bfloat16()
    dout_cpu = torch.randn(s, c, generator=g).bfloat16()
    first = None
    for it in range(40):
        junk = [torch.full((random.randint(1, 1 << 22),), float("nan"), device=DEV) for _ in range(random.randint(0, 3))]
        qkv, dout = qkv_cpu.to(DEV), dout_cpu.to(DEV)
        del junk
        out, lse = ops.mhsa(qkv, 1, h, window, return_lse=True, dropout_p=p, dropout_seed=7)
        dqkv = ops.mhsa_backward(qkv, out, dout, lse, 1, h, window, dropout_p=p, dropout_seed=7)
        assert torch.isfinite(dqkv.float()).all()
        if first is None:
            first = dqkv.clone()
        else:
            assert torch.equal(dqkv, first), it


def test_transformer_model_training_step_vs_oracle_autograd(graph_o32, golden_cfg1_tfm):
    """The Transformer-processor model (mesh-node self attention): forward + backward through the nn.Module against the
    oracle under torch autograd, golden weights of the reference."""
    from test_gpu_parity import _build

    gold = golden_cfg1_tfm
    sd = split_prefix(gold, "sd.")
    graph = _f64(graph_tensors(graph_o32))
    dy = torch.randn(gold["y"].shape, generator=torch.Generator().manual_seed(4))
    want, rsd = _oracle_grads(lambda s_, xx: ref.model_forward(s_, graph, xx, processor="Transformer", **KW), sd, gold["x"], dy)
    assert rel_err(want, gold["y"]) < 1e-4
    model, _ = _build(graph_o32, 64, 4, processor="Transformer")
    model.load_state_dict(sd)
    model = model.to(DEV)
    y = model(gold["x"].to(DEV))
    assert y.requires_grad and rel_err(y.detach(), gold["y"]) < 1e-4
    y.backward(dy.to(DEV))
    used = _compare_grads(model, rsd)
    assert any("attention.lin_qkv" in k for k in used) and any("attention.projection" in k for k in used)


def _dropout_keep_mask(seed: int, p: float, b: int, h: int, s: int, h0: int = 0, h_total: int = 0) -> torch.Tensor:
    """The kernels' counter-based keep mask (csrc/attention.hip::dropout_keep) restated with torch integer arithmetic:
    [B, H, S, S] of 0 / 1.  One 32-bit hash of (row, key >> 1) -- one multiply, two fold-downs -- decides a key pair, 15 bits
    per key."""
    m32 = 0xFFFFFFFF
    h_total = h_total or h
    bb = torch.arange(b, dtype=torch.int64).view(b, 1, 1, 1)
    hh = torch.arange(h, dtype=torch.int64).view(1, h, 1, 1) + h0
    qq = torch.arange(s, dtype=torch.int64).view(1, 1, s, 1)
    row = (bb * h_total + hh) * s + qq  # (b * H_total + h0 + h) * S + i
    col = torch.arange(s, dtype=torch.int64).view(1, 1, 1, s)
    x = ((row & m32) * 0x9E3779B1) & m32
    x = x ^ (((row >> 32) * 0x85EBCA77) & m32) ^ (seed & m32)
    x = x ^ (((col >> 1) * 0xC2B2AE3D) & m32)
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & m32
    x = x ^ (x >> 15)
    bits = (x >> (16 * (col & 1))) & 0x7FFF
    thr = min(int(p * 32768.0 + 0.5), 32768)
    return (bits >= thr).to(torch.float64) if p < 1.0 else torch.zeros(b, h, s, s, dtype=torch.float64)


@pytest.mark.parametrize("processor", ["Transformer", "GraphTransformer"])
def test_bf16_model_training_step_on_assembled_input_rows(graph_o32, monkeypatch, processor):
    """bf16 training with an input width that is a whole number of the GEMM's K multiples (2 x 26 variables + 12 node
    attributes = 64, as config 3's 192): the model input is assembled once as ``[features | 1 | 0-pad]`` rows
    (``training._AssembleNodes``: 128 columns here), wider than the embedding Linears' K, which accept that width only when
    told (``padded_input``).  The Transformer processor keeps the external mesh order and used to reach the mappers through
    their module-level call, which cannot be told: at config 3's size that route raised in round 6 (found by the
    training-step table, not by the suite, whose widths did not cross a K multiple).  Output and every gradient against the
    f32 training step of the same weights."""
    from test_gpu_parity import _build

    g = torch.Generator().manual_seed(12)
    n_grid = graph_o32["data"].num_nodes
    x = torch.randn(1, 2, 1, n_grid, 26, generator=g).to(DEV)
    res = {}
    for mode in ("fp32", "bf16"):
        monkeypatch.setenv("ANEMOI_AMD_DTYPE", mode)
        torch.manual_seed(77)
        model, _ = _build(graph_o32, 64, 2, processor=processor, n_prog=20, n_forc=6, n_diag=1)
        assert model.encoder.emb_nodes_src.in_features == 64
        with torch.no_grad():
            for name, p in model.named_parameters():
                if name.endswith("trainable"):
                    p.normal_(0.0, 0.1)
        model = model.to(DEV).train()
        for m in model.modules():
            if hasattr(m, "dropout_p"):
                m.dropout_p = 0.0
        y = model(x)
        if mode == "fp32":
            dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(2)).to(DEV)
        y.backward(dy)
        res[mode] = (y.detach().float().clone(), {k: p.grad.float().clone() for k, p in model.named_parameters()
                                                   if p.grad is not None})
    assert rel_err(res["bf16"][0], res["fp32"][0]) < 5e-2
    assert set(res["bf16"][1]) == set(res["fp32"][1])
    scale_all = max(float(g_.abs().max()) for g_ in res["fp32"][1].values())
    for k, g32 in res["fp32"][1].items():
        err = float((res["bf16"][1][k] - g32).abs().max())
        assert err <= 8e-2 * max(float(g32.abs().max()), 0.05 * scale_all), (k, err, float(g32.abs().max()))


@pytest.mark.parametrize("dtype,b,s,h,d,window,p", [
    (torch.float32, 2, 150, 3, 5, -1, 0.3), (torch.float32, 1, 97, 2, 16, 20, 0.5), (torch.bfloat16, 1, 200, 4, 64, -1, 0.1),
    (torch.float32, 1, 64, 2, 8, -1, 1.0), (torch.bfloat16, 2, 130, 4, 32, -1, 0.25),
    # the MFMA kernels (bf16, D = 64 / 32): whole 512-query blocks + a ragged one, the key-split rows behind the last
    # block (S = 514), a sliding window, D = 32
    (torch.bfloat16, 1, 700, 4, 64, -1, 0.1), (torch.bfloat16, 1, 514, 2, 64, -1, 0.3), (torch.bfloat16, 2, 700, 4, 32, -1, 0.25),
    (torch.bfloat16, 1, 1100, 2, 64, 70, 0.2),
])
def test_mhsa_attention_dropout_forward_and_backward(dtype, b, s, h, d, window, p):
    """Attention dropout (reference layers/attention.py:90-105, training mode): forward and gradients against torch
    autograd in f64 through dropout(softmax(.)) V with the SAME mask (the kernels' hash restated above); the fraction of
    kept probabilities matches 1 - p; p = 1 drops everything."""
    from anemoi_models_amd import autograd

    g = torch.Generator().manual_seed(s + d)
    seed = 123456789 + s
    c = h * d
    qkv = (torch.randn(b * s, 3 * c, generator=g) * 0.8).to(dtype)
    dout = torch.randn(b * s, c, generator=g).to(dtype)
    keep = _dropout_keep_mask(seed, p, b, h, s)
    if 0.0 < p < 1.0:
        assert abs(float(keep.mean()) - (1.0 - p)) < 0.02
    ref_in = qkv.double().requires_grad_()
    q, k, v = (t.reshape(b, s, h, d).permute(0, 2, 1, 3) for t in ref_in.split(c, dim=1))
    sc = q @ k.transpose(-1, -2) / d**0.5
    if window >= 0:
        i = torch.arange(s)
        sc = sc.masked_fill((i[:, None] - i[None, :]).abs() > window, float("-inf"))
    prob = torch.softmax(sc, -1) * keep * (1.0 / (1.0 - p) if p < 1.0 else 0.0)
    want = (prob @ v).permute(0, 2, 1, 3).reshape(b * s, c)
    want.backward(dout.double())
    x = qkv.to(DEV).requires_grad_()
    got = autograd.mhsa(x, b, h, window, p, seed)
    got.backward(dout.to(DEV))
    if p >= 1.0:
        assert not got.detach().any() and not x.grad.any()
        return
    assert rel_err(got.detach(), want.detach()) < (2e-5 if dtype == torch.float32 else 2e-2)
    assert rel_err(x.grad, ref_in.grad) < (1e-4 if dtype == torch.float32 else 3e-2)
    # the same seed gives the same mask, another seed another one
    again = autograd.mhsa(qkv.to(DEV), b, h, window, p, seed)
    other = autograd.mhsa(qkv.to(DEV), b, h, window, p, seed + 1)
    assert torch.equal(again, got.detach()) and not torch.equal(other, got.detach())


def test_mhsa_attention_dropout_on_the_mfma_kernels_at_mesh_size():
    """Attention dropout at the sequence length of BASELINE config 2's mesh (S = 10 242 = 20 blocks of 512 queries + 2
    key-split rows, heads of 64, bf16: the MFMA forward / dK dV / dQ kernels with the mask applied to their packed
    probabilities) against torch autograd in f64 with the restated mask; the reference's constructor default p = 0.1
    (layers/processor.py:99)."""
    from anemoi_models_amd import autograd

    b, s, h, d, p = 1, 10242, 2, 64, 0.1
    g = torch.Generator().manual_seed(5)
    seed, c = 20260101, h * d
    qkv = (torch.randn(b * s, 3 * c, generator=g) * 0.8).bfloat16()
    dout = torch.randn(b * s, c, generator=g).bfloat16()
    keep = _dropout_keep_mask(seed, p, b, h, s)
    assert abs(float(keep.mean()) - (1.0 - p)) < 2e-3
    ref_in = qkv.double().requires_grad_()
    q, k, v = (t.reshape(b, s, h, d).permute(0, 2, 1, 3) for t in ref_in.split(c, dim=1))
    prob = torch.softmax(q @ k.transpose(-1, -2) / d**0.5, -1) * keep * (1.0 / (1.0 - p))
    want = (prob @ v).permute(0, 2, 1, 3).reshape(b * s, c)
    want.backward(dout.double())
    x = qkv.to(DEV).requires_grad_()
    got = autograd.mhsa(x, b, h, -1, p, seed)
    got.backward(dout.to(DEV))
    e_out, e_grad = rel_err(got.detach(), want.detach()), rel_err(x.grad, ref_in.grad)
    print(f"MFMA attention dropout p = {p} at S = {s}: output rel err {e_out:.3e}, d qkv rel err {e_grad:.3e}")
    assert e_out < 2e-2 and e_grad < 3e-2
    same_bits_or_last_bit_rows(autograd.mhsa(qkv.to(DEV), b, h, -1, p, seed), got.detach(), "two attention forwards with dropout")


@pytest.mark.parametrize("batch_size,num_heads,mult,p", [(3, 4, 5, 0.4), (8, 1, 10, 0.0), (2, 20, 1, 1.0), (5, 7, 3, 0.73)])
def test_multi_head_self_attention_module_like_the_reference_tests(batch_size, num_heads, mult, p):
    """reference tests/layers/test_attention.py:38-78: a MultiHeadSelfAttention in its default (training) mode with an
    arbitrary dropout_p and small odd head sizes: forward shape, backward to the input."""
    from anemoi_models_amd.layers.attention import MultiHeadSelfAttention

    embed_dim = num_heads * mult
    torch.manual_seed(7)
    mhsa = MultiHeadSelfAttention(num_heads, embed_dim, dropout_p=p).to(DEV)
    assert mhsa.training and mhsa.dropout_p == p
    x = torch.randn(batch_size * 2, embed_dim, device=DEV, requires_grad=True)
    out = mhsa.forward(x, [list(x.shape)], batch_size)
    assert out.shape == x.shape
    out.sum().backward()
    assert x.grad is not None and x.grad.shape == x.shape and bool(torch.isfinite(x.grad).all())
    with torch.no_grad():  # inference route of a module left in training mode: dropout still applies, eval() removes it
        y_train = mhsa(x.detach(), [list(x.shape)], batch_size)
        mhsa.eval()
        y_eval, y_eval2 = mhsa(x.detach(), [list(x.shape)], batch_size), mhsa(x.detach(), [list(x.shape)], batch_size)
    assert torch.equal(y_eval, y_eval2)
    if 0.0 < p < 1.0:
        assert not torch.equal(y_train, y_eval)


@pytest.mark.parametrize("dtype,tol,c,h,edge_dim", [
    (torch.float32, 2e-3, 128, 8, 39), (torch.bfloat16, 8e-2, 128, 8, 39),
    # shapes the FOLDED edge kernels do not come in (autograd.folded_edge_route; they raised in training mode until round 6):
    # heads of 12 (three lanes), 10 and 6 (not a multiple of 4: zero-padded heads), 48 (six bf16 lanes), a single bf16 head
    # (its packed u / t columns would break the 16-byte row alignment)
    (torch.float32, 2e-3, 96, 8, 11), (torch.bfloat16, 8e-2, 192, 16, 11), (torch.float32, 2e-3, 160, 16, 11),
    (torch.float32, 2e-3, 96, 16, 3), (torch.bfloat16, 8e-2, 192, 4, 3), (torch.bfloat16, 8e-2, 64, 1, 3),
])
def test_blocks_outside_the_folded_edge_kernels_train_through_the_explicit_conv(dtype, tol, c, h, edge_dim, monkeypatch):
    """edge_dim = 39 (the reference's own mapper tests use 1 + 32 attributes + 6 trainable) is beyond the folded edge
    kernels' width, and so are some head sizes and head counts: the differentiable blocks then run lin_edge as a GEMM and
    the conv on explicit per-edge features (anemoi_gt_conv + its backward kernels, heads zero-padded where needed).
    Forward and every gradient against the oracle's autograd; the inference route agrees with the training route."""
    from anemoi_models_amd import autograd
    from anemoi_models_amd.layers.block import GraphTransformerMapperBlock, GraphTransformerProcessorBlock

    monkeypatch.setenv("ANEMOI_AMD_DTYPE", "fp32" if dtype == torch.float32 else "bf16")
    assert not autograd.folded_edge_route(dtype, c, h, (edge_dim + 1 + 3) // 4 * 4)
    g = torch.Generator().manual_seed(39)
    n, n_src, e = 150, 210, 1200
    torch.manual_seed(5)
    # --- processor block
    blk = GraphTransformerProcessorBlock(c, 2 * c, c, edge_dim=edge_dim, num_heads=h)
    ei = torch.stack([torch.randint(0, n, (e,), generator=g), torch.randint(0, n - 1, (e,), generator=g)])
    x0, ea0 = torch.randn(n, c, generator=g), torch.randn(e, edge_dim, generator=g)
    rsd = {"x." + k: v.detach().double().requires_grad_() for k, v in blk.named_parameters()}
    xr, ear = x0.double().requires_grad_(), ea0.double().requires_grad_()
    want = ref.gt_processor_block(rsd, "x", xr, ear, ei, h)
    want.sum().backward()
    blk = blk.to(DEV)
    x, ea = x0.to(DEV).requires_grad_(), ea0.to(DEV).requires_grad_()
    y, _ = blk(x, ea, ei.to(DEV), None, 1)
    assert rel_err(y.detach(), want.detach()) < tol
    y.sum().backward()
    assert rel_err(x.grad, xr.grad) < tol and rel_err(ea.grad, ear.grad) < tol
    _compare_block_grads(blk, rsd, tol=tol)
    with torch.no_grad():
        y_inf, _ = blk(x.detach(), ea.detach(), ei.to(DEV), None, 1)
    assert rel_err(y_inf, want.detach()) < tol
    # --- mapper block
    blk = GraphTransformerMapperBlock(c, 2 * c, c, edge_dim=edge_dim, num_heads=h)
    ei = torch.stack([torch.randint(0, n_src, (e,), generator=g), torch.randint(0, n - 1, (e,), generator=g)])
    xs0, xd0 = torch.randn(n_src, c, generator=g), torch.randn(n, c, generator=g)
    rsd = {"x." + k: v.detach().double().requires_grad_() for k, v in blk.named_parameters()}
    xsr, xdr, ear = xs0.double().requires_grad_(), xd0.double().requires_grad_(), ea0.double().requires_grad_()
    want = ref.gt_mapper_block(rsd, "x", xsr, xdr, ear, ei, h)
    want.sum().backward()
    blk = blk.to(DEV)
    xs, xd, ea = xs0.to(DEV).requires_grad_(), xd0.to(DEV).requires_grad_(), ea0.to(DEV).requires_grad_()
    (_, y), _ = blk((xs, xd), ea, ei.to(DEV), None, 1, size=(n_src, n))
    assert rel_err(y.detach(), want.detach()) < tol
    y.sum().backward()
    assert rel_err(xs.grad, xsr.grad) < tol and rel_err(xd.grad, xdr.grad) < tol and rel_err(ea.grad, ear.grad) < tol
    _compare_block_grads(blk, rsd, tol=tol)


@pytest.mark.parametrize("mode,c,h,edge_dim", [("fp32", 64, 16, 3), ("bf16", 128, 16, 11), ("fp32", 96, 8, 11), ("bf16", 192, 4, 3),
                                               ("fp32", 128, 16, 23), ("fp32", 32, 32, 4)])
def test_blocks_train_on_an_edge_set_without_edges(mode, c, h, edge_dim, monkeypatch):
    """E = 0 (PyG's propagate takes an empty edge_index; the op fuzzer found the differentiable blocks' explicit-conv route
    raising on it in round 6 -- empty GEMM operands, an ambiguous reshape): processor and mapper block, folded and explicit
    routes, head size 1 included: the output is the oracle's, the backward runs and every gradient it returns is the oracle's."""
    from anemoi_models_amd.layers.block import GraphTransformerMapperBlock, GraphTransformerProcessorBlock

    monkeypatch.setenv("ANEMOI_AMD_DTYPE", mode)
    tol = 2e-3 if mode == "fp32" else 8e-2
    g = torch.Generator().manual_seed(c + h)
    n, n_src = 70, 45
    ei = torch.zeros(2, 0, dtype=torch.int64)
    ea0 = torch.zeros(0, edge_dim)
    torch.manual_seed(3)
    for mapper in (False, True):
        if mapper:
            blk = GraphTransformerMapperBlock(c, 2 * c, c, edge_dim=edge_dim, num_heads=h)
            xs0, xd0 = torch.randn(n_src, c, generator=g), torch.randn(n, c, generator=g)
            rsd = {"x." + k: v.detach().double().requires_grad_() for k, v in blk.named_parameters()}
            ins_r = [xs0.double().requires_grad_(), xd0.double().requires_grad_()]
            want = ref.gt_mapper_block(rsd, "x", ins_r[0], ins_r[1], ea0.double(), ei, h)
            blk = blk.to(DEV)
            ins = [t.to(DEV).requires_grad_() for t in (xs0, xd0)]
            (_, y), _ = blk((ins[0], ins[1]), ea0.to(DEV), ei.to(DEV), None, 1, size=(n_src, n))
        else:
            blk = GraphTransformerProcessorBlock(c, 2 * c, c, edge_dim=edge_dim, num_heads=h)
            x0 = torch.randn(n, c, generator=g)
            rsd = {"x." + k: v.detach().double().requires_grad_() for k, v in blk.named_parameters()}
            ins_r = [x0.double().requires_grad_()]
            want = ref.gt_processor_block(rsd, "x", ins_r[0], ea0.double(), ei, h)
            blk = blk.to(DEV)
            ins = [x0.to(DEV).requires_grad_()]
            y, _ = blk(ins[0], ea0.to(DEV), ei.to(DEV), None, 1)
        assert rel_err(y.detach(), want.detach()) < tol
        want.sum().backward()
        y.float().sum().backward()
        assert rel_err(ins[-1].grad, ins_r[-1].grad) < tol  # the destination rows' gradient
        scale = max(float(v.grad.abs().max()) for v in rsd.values() if v.grad is not None)
        for k, p in blk.named_parameters():
            want_g = rsd["x." + k].grad
            if p.grad is None:  # nothing reached this parameter: the oracle's gradient is zero (keys / values / lin_edge)
                assert want_g is None or float(want_g.abs().max()) <= 1e-12 * scale, k
            else:
                assert float((p.grad.cpu() - want_g.float()).abs().max()) <= tol * max(float(want_g.abs().max()), 0.02 * scale), k


@pytest.mark.parametrize("mode,c,h,edge_dim", [("bf16", 192, 2, 4), ("bf16", 128, 1, 23), ("bf16", 256, 2, 23), ("fp32", 96, 8, 11)])
def test_graph_transformer_block_inference_at_heads_the_edge_kernels_do_not_take(mode, c, h, edge_dim, monkeypatch):
    """The INFERENCE route of a block at head sizes beyond the generic edge kernel's 64 channels that the fast kernels do not
    take either (bf16 heads of 96; heads of 128 with 23 edge attributes) used to raise from the kernel dispatch (found by the
    block fuzzer in round 6); it now runs lin_edge as a GEMM and the conv on explicit edge features, as the training route
    does.  Against the oracle; the last case (heads of 12, the generic kernel) is the control."""
    from anemoi_models_amd.layers.block import GraphTransformerProcessorBlock

    monkeypatch.setenv("ANEMOI_AMD_DTYPE", mode)
    g = torch.Generator().manual_seed(c + edge_dim)
    n, e = 150, 1800
    torch.manual_seed(2)
    blk = GraphTransformerProcessorBlock(c, 2 * c, c, edge_dim=edge_dim, num_heads=h)
    ei = torch.stack([torch.randint(0, n, (e,), generator=g), torch.randint(0, n - 1, (e,), generator=g)])
    x0, ea0 = torch.randn(n, c, generator=g), torch.randn(e, edge_dim, generator=g)
    sd = {"x." + k: v.detach().double() for k, v in blk.named_parameters()}
    want = ref.gt_processor_block(sd, "x", x0.double(), ea0.double(), ei, h)
    blk = blk.to(DEV).eval()
    with torch.no_grad():
        y, _ = blk(x0.to(DEV), ea0.to(DEV), ei.to(DEV), None, 1)
    assert rel_err(y, want) < (2e-3 if mode == "fp32" else 8e-2)


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("train", [False, True])
def test_gnn_blocks_on_an_edge_set_without_edges(mode, train, monkeypatch):
    """GraphConvProcessorBlock / GraphConvMapperBlock with E = 0 (reference layers/block.py:193-286 through PyG's propagate on
    an empty edge_index): empty new edge state, the node MLP sees zero sums.  Inference and training route against the oracle
    (both raised on the empty operands until the block fuzzer of round 6)."""
    from anemoi_models_amd.layers.block import GraphConvMapperBlock, GraphConvProcessorBlock

    monkeypatch.setenv("ANEMOI_AMD_DTYPE", mode)
    tol = 2e-3 if mode == "fp32" else 8e-2
    c, n, n_src = 128, 60, 45
    g = torch.Generator().manual_seed(8)
    ei, ea0 = torch.zeros(2, 0, dtype=torch.int64), torch.zeros(0, c)
    torch.manual_seed(4)
    x0, xs0 = torch.randn(n, c, generator=g), torch.randn(n_src, c, generator=g)
    for blk, is_mapper in ((GraphConvProcessorBlock(c, c), False), (GraphConvMapperBlock(c, c, update_src_nodes=True), True)):
        sd = {"x." + k: v.detach().double() for k, v in blk.named_parameters()}
        if is_mapper:
            (ws, wd), we = ref.gnn_mapper_block(sd, "x", xs0.double(), x0.double(), ea0.double(), ei, True)
        else:
            wd, we = ref.gnn_processor_block(sd, "x", x0.double(), ea0.double(), ei)
        blk = blk.to(DEV).train(train)
        xd, xs = x0.to(DEV).requires_grad_(train), xs0.to(DEV).requires_grad_(train)
        with torch.enable_grad() if train else torch.no_grad():
            if is_mapper:
                (ys, yd), ye = blk((xs, xd), ea0.to(DEV), ei.to(DEV), (None, None, None), size=(n_src, n))
                assert rel_err(ys.detach(), ws) < tol
            else:
                yd, ye = blk(xd, ea0.to(DEV), ei.to(DEV), (None, None, None))
            assert rel_err(yd.detach(), wd) < tol and ye.shape == we.shape == (0, c)
            if train:
                yd.float().sum().backward()
                assert xd.grad is not None and bool(torch.isfinite(xd.grad).all())


@pytest.mark.parametrize("pair", [False, True])
def test_graph_conv_module_forward_and_backward(golden_blocks, pair):
    """``GraphConv.forward`` on its own (reference layers/conv.py:62-76): ``edges_new = edge_mlp(cat[x_i, x_j, e]) + e``
    in the caller's edge order and its sum over the destinations, outputs and all gradients against plain torch f64."""
    from anemoi_models_amd.layers.conv import GraphConv

    b = golden_blocks
    sd = {k[len("conv."):]: v for k, v in split_prefix(b, "gnn.sd.").items() if k.startswith("conv.")}
    conv = GraphConv(64, 64, mlp_extra_layers=0, activation="SiLU")
    conv.load_state_dict(sd)
    conv = conv.to(DEV)
    ei = b["gnn.edge_index"]
    n = b["gnn.x"].shape[0]
    torch.manual_seed(3)
    x_src = b["gnn.x"].clone()
    x_dst = torch.randn(n, 64) if pair else x_src
    e = b["gnn.edge_attr"].clone()
    xs, xd, eg = x_src.to(DEV).requires_grad_(), x_dst.to(DEV).requires_grad_(), e.to(DEV).requires_grad_()
    out, edges_new = conv((xs, xd) if pair else xs, eg, ei.to(DEV), size=(n, n) if pair else None)
    w_out, w_e = torch.randn(n, 64), torch.randn(ei.shape[1], 64)
    ((out * w_out.to(DEV)).sum() + (edges_new * w_e.to(DEV)).sum()).backward()

    rsd = {k: v.double().requires_grad_() for k, v in sd.items()}
    xs_r, xd_r, e_r = x_src.double().requires_grad_(), x_dst.double().requires_grad_(), e.double().requires_grad_()
    x_j, x_i = xs_r[ei[0]], (xd_r if pair else xs_r)[ei[1]]
    F = torch.nn.functional
    h = F.silu(F.linear(torch.cat([x_i, x_j, e_r], 1), rsd["edge_mlp.model.0.weight"], rsd["edge_mlp.model.0.bias"]))
    h = F.silu(F.linear(h, rsd["edge_mlp.model.2.weight"], rsd["edge_mlp.model.2.bias"]))
    h = F.linear(h, rsd["edge_mlp.model.4.weight"], rsd["edge_mlp.model.4.bias"])
    h = F.layer_norm(h, (64,), rsd["edge_mlp.model.5.weight"], rsd["edge_mlp.model.5.bias"], 1e-5)
    en_r = h + e_r
    out_r = torch.zeros(n, 64, dtype=torch.float64).index_add(0, ei[1], en_r)
    ((out_r * w_out.double()).sum() + (en_r * w_e.double()).sum()).backward()

    assert rel_err(edges_new.detach(), en_r.detach()) < 1e-4 and rel_err(out.detach(), out_r.detach()) < 1e-4
    assert rel_err(eg.grad, e_r.grad) < 2e-3 and rel_err(xs.grad, xs_r.grad) < 2e-3
    if pair:
        assert rel_err(xd.grad, xd_r.grad) < 2e-3
    for k, p in conv.named_parameters():
        assert rel_err(p.grad, rsd[k].grad) < 5e-3, k
    with torch.no_grad():  # the inference route gives the same numbers
        out2, en2 = conv((xs, xd) if pair else xs, eg, ei.to(DEV))
    assert rel_err(out2, out.detach()) < 1e-5 and rel_err(en2, edges_new.detach()) < 1e-5


def test_module_level_model_groups_ranks_sharing_one_gpu(tmp_path):
    """The reference's module-level calls with a model group on the HIP kernels (reference layers/processor.py:103-343,
    layers/mapper.py:239-418): GraphTransformer / GNN / Transformer processors forward AND backward, GraphTransformer
    mappers forward, two ranks sharing cuda:0 (gloo through host memory) against the unsharded modules; the parameter
    gradients summed over the ranks equal the unsharded gradients."""
    import os
    import subprocess
    import sys

    port = 29700 + (os.getpid() % 150)
    out = str(tmp_path / "res")
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_gpu_shared_modules.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", str(port), out]) for r in range(2)]
    try:
        codes = [p.wait(timeout=900) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
    assert codes == [0, 0]
    for r in range(2):
        info = torch.load(f"{out}.{r}")
        assert len(info) == 10, sorted(info)
        assert info.pop("tfm_dropout.acts") == 1.0  # the train-mode call really dropped probabilities
        for k, v in info.items():
            assert v < (2e-5 if k.endswith(".fwd") else 2e-4), (r, k, v)
