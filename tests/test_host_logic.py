"""CPU tests of the host side: state_dict layout, config instantiation, CSR plans, launch-sequence wiring.

The kernels themselves are GPU-only; here ``anemoi_models_amd.ops`` is replaced by oracle-backed torch functions
(tests/_cpu_ops.py) so that the orchestration in the layer mirrors can be compared with the golden vectors.
"""

import json
import os

import pytest
import torch

import _cpu_ops
from conftest import GOLDEN
from conftest import split_prefix
from anemoi_models_amd import runtime
from anemoi_models_amd.models import AnemoiModelEncProcDec
from anemoi_models_amd.utils.indices import SimpleDataIndices
from anemoi_models_amd.utils.presets import hierarchical_model_config
from anemoi_models_amd.utils.presets import model_config


def build_model(graph, processor="GraphTransformer", mappers="GraphTransformer"):
    idx = SimpleDataIndices(n_prognostic=10, n_forcing=2, n_diagnostic=1)
    return AnemoiModelEncProcDec(model_config=model_config(processor, 64, 4, 16, mappers=mappers), data_indices=idx,
                                 graph_data=graph)


@pytest.mark.parametrize("processor", ["GraphTransformer", "GNN", "Transformer", "GNN_all"])
def test_state_dict_layout_matches_reference(graph_o32, processor):
    with open(os.path.join(GOLDEN, "state_dict_keys.json")) as f:
        ref = json.load(f)[processor]
    model = build_model(graph_o32, "GNN", "GNN") if processor == "GNN_all" else build_model(graph_o32, processor)
    sd = model.state_dict()
    assert list(sd.keys()) == list(ref.keys()) or set(sd.keys()) == set(ref.keys())
    assert {k: list(v.shape) for k, v in sd.items()} == ref


def test_same_seed_same_initial_weights_as_reference(graph_o32, golden_cfg1_gt):
    """Parameters are created in the reference's order, so a seeded construction gives the reference's init."""
    torch.manual_seed(1234)
    sd = build_model(graph_o32).state_dict()
    gold = split_prefix(golden_cfg1_gt, "sd.")
    for k in ("encoder.emb_nodes_src.weight", "processor.proc.1.blocks.1.lin_value.weight",
              "decoder.node_data_extractor.1.weight", "decoder.proc.node_dst_mlp.3.bias"):
        assert torch.equal(sd[k], gold[k]), k


def test_edge_plan_is_stable_dst_sort():
    g = torch.Generator().manual_seed(0)
    ei = torch.stack([torch.randint(0, 50, (400,), generator=g), torch.randint(0, 30, (400,), generator=g)])
    plan = runtime.build_edge_plan(ei, 50, 30)
    assert plan.rowptr.dtype == torch.int32 and plan.col.dtype == torch.int32
    assert plan.rowptr[0] == 0 and plan.rowptr[-1] == 400
    dst_sorted = ei[1][plan.perm.long()]
    assert torch.all(dst_sorted[1:] >= dst_sorted[:-1])
    for d in range(30):
        seg = plan.perm[plan.rowptr[d]:plan.rowptr[d + 1]].long()
        assert torch.all(ei[1][seg] == d)
        assert torch.all(seg[1:] > seg[:-1])  # stable: original edge order inside a destination
        assert torch.equal(plan.col[plan.rowptr[d]:plan.rowptr[d + 1]].long(), ei[0][seg])
    with pytest.raises(ValueError):
        runtime.build_edge_plan(ei, 50, 29)


def test_expand_edges_bit_exact(golden_index_ops):
    z = golden_index_ops
    inc = torch.tensor([[70], [31]], dtype=torch.int64)
    assert torch.equal(runtime.expand_edges(z["expand.edge_index"], inc, 3), z["expand.out"])


def test_forward_with_gradients_enabled(graph_o32, golden_cfg1_gt, monkeypatch):
    """With autograd on, the flat GraphTransformer model takes the differentiable route (autograd.model_forward: same
    result as the inference route, an autograd graph behind it)."""
    _cpu_ops.install(monkeypatch)
    model = build_model(graph_o32)
    model.load_state_dict(split_prefix(golden_cfg1_gt, "sd."))
    y = model(golden_cfg1_gt["x"])
    assert y.requires_grad and y.grad_fn is not None
    torch.testing.assert_close(y.detach(), golden_cfg1_gt["y"], atol=5e-4, rtol=5e-4)
    with torch.no_grad():
        torch.testing.assert_close(model(golden_cfg1_gt["x"]), y.detach(), atol=1e-4, rtol=1e-4)
    y2 = model(golden_cfg1_gt["x"].repeat(2, 1, 1, 1, 1))  # batch 2: the batched graph, both samples alike
    torch.testing.assert_close(y2[0].detach(), y[0].detach(), atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(y2[1].detach(), y[0].detach(), atol=1e-4, rtol=1e-4)


def test_kernels_refuse_cpu_tensors():
    from anemoi_models_amd import ops

    with pytest.raises(RuntimeError):
        ops.layer_norm(torch.zeros(4, 8), torch.ones(8), torch.zeros(8))


def test_gt_model_wiring_matches_golden(graph_o32, golden_cfg1_gt, monkeypatch):
    _cpu_ops.install(monkeypatch)
    gold = golden_cfg1_gt
    model = build_model(graph_o32)
    model.load_state_dict(split_prefix(gold, "sd."))
    model.eval()
    with torch.no_grad():
        y = model(gold["x"])
    torch.testing.assert_close(y, gold["y"], atol=1e-4, rtol=1e-4)
    # the gathered edge attributes are kept between calls (runtime.edge_attr_csr_cached) -- and dropped when the trainable
    # edge tensor or the attribute buffer is written to in place (an optimiser step, a loaded checkpoint)
    calls = []
    real = _cpu_ops.edge_attr_csr
    monkeypatch.setattr("anemoi_models_amd.ops.edge_attr_csr", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    with torch.no_grad():
        y2 = model(gold["x"])
        assert not calls and torch.equal(y2, y)
        model.processor.trainable.trainable.add_(0.5)
        y3 = model(gold["x"])
        assert len(calls) == 1 and not torch.allclose(y3, y)
        model.processor.trainable.trainable.sub_(0.5)
        torch.testing.assert_close(model(gold["x"]), y, atol=1e-5, rtol=1e-5)


def test_layer_norm_fold_wiring_bf16(graph_o32, golden_cfg1_gt, monkeypatch):
    """bf16 route with LayerNorm folded into the consuming Linear (row_stats + linear(ln=...)) against the same model
    with the fold switched off and against the f32 golden output (host logic only; kernels are the CPU stand-ins)."""
    _cpu_ops.install(monkeypatch)
    gold = golden_cfg1_gt
    model = build_model(graph_o32)
    model.load_state_dict(split_prefix(gold, "sd."))
    model.eval()
    monkeypatch.setenv("ANEMOI_AMD_DTYPE", "bf16")
    outs = {}
    for fold in ("1", "0"):
        monkeypatch.setenv("ANEMOI_AMD_LN_FOLD", fold)
        with torch.no_grad():
            outs[fold] = model(gold["x"])
    scale = float(gold["y"].abs().max())
    assert float((outs["1"] - outs["0"]).abs().max()) < 0.03 * scale
    assert float((outs["1"] - gold["y"]).abs().max()) < 0.05 * scale


def test_gt_blocks_wiring_matches_golden(golden_blocks, monkeypatch):
    from anemoi_models_amd.layers.block import GraphTransformerMapperBlock
    from anemoi_models_amd.layers.block import GraphTransformerProcessorBlock

    _cpu_ops.install(monkeypatch)
    b = golden_blocks
    blk = GraphTransformerProcessorBlock(128, 512, 128, edge_dim=11, num_heads=16, activation="GELU")
    blk.load_state_dict(split_prefix(b, "gtp.sd."))
    with torch.no_grad():
        y, ea = blk(b["gtp.x"], b["gtp.edge_attr"], b["gtp.edge_index"], (None, None, None), 1)
    torch.testing.assert_close(y, b["gtp.y"], atol=2e-5, rtol=2e-5)
    assert ea is b["gtp.edge_attr"]

    mb = GraphTransformerMapperBlock(64, 256, 64, edge_dim=11, num_heads=16, activation="GELU").eval()
    mb.load_state_dict(split_prefix(b, "gtm.sd."))
    with torch.no_grad():
        (ys, yd), _ = mb((b["gtm.x_src"], b["gtm.x_dst"]), b["gtm.edge_attr"], b["gtm.edge_index"],
                         (None, None, None), 1, size=(180, 90))
        assert ys is b["gtm.x_src"]
        torch.testing.assert_close(yd, b["gtm.y_dst"], atol=2e-5, rtol=2e-5)
        # reference test_GraphTransformerMapperBlock_chunking: chunked == unchunked
        monkeypatch.setenv("ANEMOI_INFERENCE_NUM_CHUNKS", "5")
        (_, yc), _ = mb((b["gtm.x_src"], b["gtm.x_dst"]), b["gtm.edge_attr"], b["gtm.edge_index"],
                        (None, None, None), 1, size=(180, 90))
        assert torch.allclose(yd, yc, atol=1e-4)
        with pytest.raises(ValueError):
            mb((b["gtm.x_src"], b["gtm.x_dst"]), b["gtm.edge_attr"], b["gtm.edge_index"], (None,) * 3, 1,
               size=(180, 91))


def test_bad_activation_raises_runtime_error():
    from anemoi_models_amd.layers.block import GraphTransformerProcessorBlock

    with pytest.raises(RuntimeError):
        GraphTransformerProcessorBlock(64, 256, 64, edge_dim=11, num_heads=16, activation="NoSuchAct")


def test_gnn_model_and_block_wiring_match_golden(graph_o32, golden_cfg1_gnn, golden_blocks, monkeypatch):
    from anemoi_models_amd.layers.block import GraphConvProcessorBlock

    _cpu_ops.install(monkeypatch)
    b = golden_blocks
    blk = GraphConvProcessorBlock(64, 64, mlp_extra_layers=0, activation="SiLU").eval()
    blk.load_state_dict(split_prefix(b, "gnn.sd."))
    with torch.no_grad():
        y, e_new = blk(b["gnn.x"], b["gnn.edge_attr"], b["gnn.edge_index"], (None, None), None)
    torch.testing.assert_close(y, b["gnn.y"], atol=2e-5, rtol=2e-5)
    torch.testing.assert_close(e_new, b["gnn.edges_new"], atol=2e-5, rtol=2e-5)  # returned in the caller's edge order

    # the conv on its own (reference layers/conv.py:62-76): (sum over destinations, new edge state)
    with torch.no_grad():
        out, edges_new = blk.conv(b["gnn.x"], b["gnn.edge_attr"], b["gnn.edge_index"])
        out_pair, _ = blk.conv((b["gnn.x"], b["gnn.x"]), b["gnn.edge_attr"], b["gnn.edge_index"],
                               size=(b["gnn.x"].shape[0],) * 2)
    torch.testing.assert_close(edges_new, b["gnn.edges_new"], atol=2e-5, rtol=2e-5)
    agg = torch.zeros_like(b["gnn.x"]).index_add_(0, b["gnn.edge_index"][1], b["gnn.edges_new"])
    torch.testing.assert_close(out, agg, atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(out_pair, out)
    with pytest.raises(ValueError):
        blk.conv(b["gnn.x"], b["gnn.edge_attr"], b["gnn.edge_index"], size=(3, 3))

    gold = golden_cfg1_gnn
    model = build_model(graph_o32, "GNN")
    model.load_state_dict(split_prefix(gold, "sd."))
    model.eval()
    with torch.no_grad():
        out = model(gold["x"])
    torch.testing.assert_close(out, gold["y"], atol=1e-4, rtol=1e-4)


def test_transformer_model_and_block_wiring_match_golden(graph_o32, golden_cfg1_tfm, golden_blocks, monkeypatch):
    from anemoi_models_amd.layers.block import TransformerProcessorBlock

    _cpu_ops.install(monkeypatch)
    b = golden_blocks
    blk = TransformerProcessorBlock(64, 256, 8, "GELU", window_size=16, dropout_p=0.0).eval()
    blk.load_state_dict(split_prefix(b, "tfm.sd."))
    with torch.no_grad():
        y = blk(b["tfm.x"], [[192, 64]], 2)
        att = blk.attention(b["tfm.x"], [[192, 64]], 2)
    torch.testing.assert_close(att, b["tfm.att"], atol=2e-5, rtol=2e-5)
    torch.testing.assert_close(y, b["tfm.y"], atol=2e-5, rtol=2e-5)

    gold = golden_cfg1_tfm
    model = build_model(graph_o32, "Transformer")
    model.load_state_dict(split_prefix(gold, "sd."))
    model.eval()
    with torch.no_grad():
        out = model(gold["x"])
    torch.testing.assert_close(out, gold["y"], atol=1e-4, rtol=1e-4)


def test_all_gnn_model_wiring_matches_golden(graph_o32, golden_cfg1_gnn_all, monkeypatch):
    _cpu_ops.install(monkeypatch)
    gold = golden_cfg1_gnn_all
    model = build_model(graph_o32, "GNN", "GNN")
    model.load_state_dict(split_prefix(gold, "sd."))
    model.eval()
    with torch.no_grad():
        out = model(gold["x"])
    torch.testing.assert_close(out, gold["y"], atol=1e-4, rtol=1e-4)

    # the reference's per-mapper entry points: prepare_edges (layers/mapper.py:485-495) and _run_mapper (models/
    # encoder_processor_decoder.py:127-165)
    dec = model.decoder
    e_attr, e_index = dec.prepare_edges((dec.edge_inc[0, 0].item(), dec.edge_inc[1, 0].item()), 2)
    n_e = dec.edge_attr.shape[0]
    assert e_attr.shape == (2 * n_e, dec.hidden_dim) and e_index.shape == (2, 2 * n_e)
    torch.testing.assert_close(e_index[:, n_e:], dec.edge_index_base + dec.edge_inc)
    torch.testing.assert_close(e_attr[:n_e], e_attr[n_e:])
    n_h, n_d = model.node_attributes.num_nodes["hidden"], model.node_attributes.num_nodes["data"]
    x_h, x_d = torch.randn(n_h, dec.hidden_dim), torch.randn(n_d, dec.hidden_dim)  # both already in the hidden space
    shapes = ([[n_h, dec.hidden_dim]], [[n_d, dec.hidden_dim]])
    with torch.no_grad():
        direct = dec((x_h, x_d), batch_size=1, shard_shapes=shapes)
        via = model._run_mapper(dec, (x_h, x_d), batch_size=1, shard_shapes=shapes)
    torch.testing.assert_close(via, direct)


def build_hierarchical(graph):
    from anemoi_models_amd.models import AnemoiModelEncProcDecHierarchical

    idx = SimpleDataIndices(n_prognostic=10, n_forcing=2, n_diagnostic=1)
    return AnemoiModelEncProcDecHierarchical(model_config=hierarchical_model_config(64, 16), data_indices=idx,
                                             graph_data=graph)


def test_hierarchical_state_dict_layout_matches_reference(graph_hier):
    with open(os.path.join(GOLDEN, "state_dict_keys.json")) as f:
        want = json.load(f)["Hierarchical"]
    got = {k: list(v.shape) for k, v in build_hierarchical(graph_hier).state_dict().items()}
    assert got == want


def test_hierarchical_model_wiring_matches_golden(graph_hier, golden_hier_gt, monkeypatch):
    """models/hierarchical.py: down / up sweeps, skip connections and channel doubling against the real reference."""
    _cpu_ops.install(monkeypatch)
    gold = golden_hier_gt
    model = build_hierarchical(graph_hier)
    model.load_state_dict(split_prefix(gold, "sd."))
    model.eval()
    with torch.no_grad():
        y = model(gold["x"])
    torch.testing.assert_close(y, gold["y"], atol=2e-4, rtol=2e-4)


NORMALIZER_METHODS = {"default": "mean-std", "min-max": ["prog_3"], "max": ["prog_4"], "std": ["prog_5"],
                      "none": ["forc_0"], "remap": {"prog_7": "prog_6"}}  # as in tests/golden/make_golden.py


def build_interface(graph, gold):
    from anemoi_models_amd.interface import AnemoiModelInterface

    cfg = model_config("GraphTransformer", 64, 4, 16)
    cfg["data"] = {"forcing": ["forc_0", "forc_1"], "diagnostic": ["diag_0"],
                   "processors": {"normalizer": {"_target_": "anemoi.models.preprocessing.normalizer.InputNormalizer",
                                                 "config": dict(NORMALIZER_METHODS)}}}
    cfg["model"]["model"] = {"_target_": "anemoi.models.models.encoder_processor_decoder.AnemoiModelEncProcDec"}
    cfg = type(cfg)(cfg)
    stats = {k: v.numpy() for k, v in split_prefix(gold, "stat.").items()}
    idx = SimpleDataIndices(n_prognostic=10, n_forcing=2, n_diagnostic=1)
    return AnemoiModelInterface(config=cfg, graph_data=graph, statistics=stats, data_indices=idx, metadata={})


def test_interface_state_dict_and_normalizer_buffers_match_reference(graph_o32, golden_interface):
    """Same state_dict layout as the reference AnemoiModelInterface, and the InputNormalizer buffers built from the
    same config / statistics are identical to the reference's (reference preprocessing/normalizer.py:44-105)."""
    with open(os.path.join(GOLDEN, "state_dict_keys.json")) as f:
        want = json.load(f)["Interface"]
    iface = build_interface(graph_o32, golden_interface)
    got = {k: list(v.shape) for k, v in iface.state_dict().items()}
    assert got == want
    sd = split_prefix(golden_interface, "sd.")
    for side in ("pre_processors", "post_processors"):
        for buf in ("_norm_mul", "_norm_add", "_input_idx", "_output_idx"):
            key = f"{side}.processors.normalizer.{buf}"
            torch.testing.assert_close(iface.state_dict()[key].double(), sd[key].double(), atol=1e-6, rtol=1e-6)


def test_interface_predict_step_matches_golden(graph_o32, golden_interface, monkeypatch):
    """predict_step: normalise -> model -> de-normalise (reference interface/__init__.py:97-123)."""
    _cpu_ops.install(monkeypatch)
    gold = golden_interface
    iface = build_interface(graph_o32, gold)
    iface.load_state_dict(split_prefix(gold, "sd."))
    iface.eval()
    y = iface.predict_step(gold["batch"])
    torch.testing.assert_close(y, gold["y"], atol=5e-4, rtol=5e-4)
    with pytest.raises(AssertionError):
        iface.predict_step(gold["batch"][0])  # 3-dimensional input: same assertion as the reference


def test_interface_rollout_matches_golden(graph_o32, golden_interface, monkeypatch):
    """rollout: n x (model, advance_input) on the device-resident normalised state; every step of the recorded vectors
    is the real reference model + normaliser (tests/golden/make_golden.py::golden_interface)."""
    _cpu_ops.install(monkeypatch)
    gold = golden_interface
    iface = build_interface(graph_o32, gold)
    iface.load_state_dict(split_prefix(gold, "sd."))
    iface.eval()
    y = iface.rollout(gold["batch"], 3, gold["rollout_forcings"])
    assert y.shape == gold["rollout_y"].shape
    torch.testing.assert_close(y, gold["rollout_y"], atol=2e-3, rtol=2e-3)
    # without forcings the last forcing values persist: step 0 is unchanged, later steps differ
    y2 = iface.rollout(gold["batch"], 2)
    torch.testing.assert_close(y2[0], y[0])
    assert (y2[1] - y[1]).abs().max() > 1e-3
    assert iface._advance_map(torch.device("cpu")).tolist() == list(range(10)) + [-2, -3]


IMPUTER_CASES = {  # as in tests/golden/make_golden.py
    "InputImputer": {"default": "none", "mean": ["y"], "maximum": ["x"], "none": ["z"], "minimum": ["q", "other"]},
    "InputImputerDefault": {"default": "minimum"},
    "ConstantImputer": {"default": "none", 0: ["x"], 3.0: ["y"], 22.7: ["z"], 10: ["q"]},
    "DynamicInputImputer": {"default": "none", "mean": ["y", "q"], "maximum": ["x"]},
    "DynamicConstantImputer": {"default": 22.7},
}


@pytest.mark.parametrize("case", sorted(IMPUTER_CASES))
def test_imputers_match_reference(case):
    """preprocessing.imputer against vectors recorded from the reference imputers on the reference IndexCollection
    (reference preprocessing/imputer.py; cases modelled on its tests/preprocessing/test_preprocessor_imputer.py)."""
    import warnings

    import numpy as np

    from anemoi_models_amd.preprocessing import imputer

    with np.load(os.path.join(GOLDEN, "imputers.npz")) as z:
        gold = {k: torch.from_numpy(z[k]) for k in z.files}
    g = split_prefix(gold, case + ".")
    stats = None if "Constant" in case else {k: v.numpy() for k, v in split_prefix(gold, "stat.").items()}
    idx = SimpleDataIndices(n_prognostic=2, n_forcing=2, n_diagnostic=1, names=["x", "y", "z", "q", "other"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        imp = getattr(imputer, case.replace("Default", ""))(config=dict(IMPUTER_CASES[case]), data_indices=idx,
                                                             statistics=stats)
    x_train = g["x_train"].clone()
    t_train = imp.transform(x_train, in_place=False)
    assert torch.equal(torch.isnan(x_train), torch.isnan(g["x_train"]))  # in_place=False leaves the input alone
    torch.testing.assert_close(t_train, g["t_train"], rtol=0, atol=0, equal_nan=True)
    torch.testing.assert_close(imp.transform(g["x_infer"], in_place=False), g["t_infer"], rtol=0, atol=0,
                               equal_nan=True)
    torch.testing.assert_close(imp.inverse_transform(g["y_train"], in_place=False), g["inv_train"], rtol=0, atol=0,
                               equal_nan=True)
    torch.testing.assert_close(imp.inverse_transform(g["y_infer"], in_place=False), g["inv_infer"], rtol=0, atol=0,
                               equal_nan=True)
    torch.testing.assert_close(imp.loss_mask_training, g["loss_mask"], rtol=0, atol=0)
    inplace = g["x_train"].clone()
    assert imp.transform(inplace) is inplace  # in place by default
    with pytest.raises(ValueError):
        imp.transform(torch.zeros(2, 7, 9))


def test_interface_predict_step_with_the_normalizer_folded_into_the_forward(graph_o32, golden_interface, monkeypatch):
    """Second and later predict_step calls hand the raw state and the InputNormalizer's affine maps to the model
    (assemble_nodes / finalize_output): same result as the reference interface."""
    _cpu_ops.install(monkeypatch)
    monkeypatch.setenv("ANEMOI_AMD_FUSE_NORMALIZER", "force")
    gold = golden_interface
    iface = build_interface(graph_o32, gold)
    iface.load_state_dict(split_prefix(gold, "sd."))
    iface.eval()
    assert iface._normalizer_affines(gold["batch"]) is None  # first call: generic route (one-off NaN check)
    y0 = iface.predict_step(gold["batch"])
    fused = iface._normalizer_affines(gold["batch"])
    assert fused is not None and fused[0][0].numel() == 12 and fused[1][0].numel() == 11
    y1 = iface.predict_step(gold["batch"])
    torch.testing.assert_close(y1, gold["y"], atol=5e-4, rtol=5e-4)
    torch.testing.assert_close(y1, y0, atol=1e-5, rtol=1e-5)
    monkeypatch.setenv("ANEMOI_AMD_FUSE_NORMALIZER", "0")
    assert iface._normalizer_affines(gold["batch"]) is None


# ------------------------------------------------------------------------------------------- a13: edge partition
def _check_khop_against_golden(z, device):
    from anemoi_models_amd.distributed import khop_edges as K

    ei, ea = z["homo.edge_index"].to(device), z["homo.edge_attr"].to(device)
    attr_l, idx_l = K.sort_edges_1hop_chunks(53, ea, ei, 5)
    assert len(attr_l) == len(idx_l) == 5
    for i in range(5):
        assert idx_l[i].dtype == torch.int64 and idx_l[i].device == ei.device
        assert torch.equal(idx_l[i].cpu(), z[f"homo.index{i}"]) and torch.equal(attr_l[i].cpu(), z[f"homo.attr{i}"])
    ei, ea = z["bip.edge_index"].to(device), z["bip.edge_attr"].to(device)
    attr_l, idx_l = K.sort_edges_1hop_chunks((70, 31), ea, ei, 4)
    for i in range(4):
        assert torch.equal(idx_l[i].cpu(), z[f"bip.index{i}"]) and torch.equal(attr_l[i].cpu(), z[f"bip.attr{i}"])
    # get_k_hop_edges on one chunk = that chunk (reference distributed/khop_edges.py:24-47)
    ei, ea = z["homo.edge_index"].to(device), z["homo.edge_attr"].to(device)
    nodes = torch.arange(53, device=device).tensor_split(5)[2]
    a, i2 = K.get_k_hop_edges(nodes, ea, ei)
    assert torch.equal(i2.cpu(), z["homo.index2"]) and torch.equal(a.cpu(), z["homo.attr2"])
    # the CSR plan of the kernels is a refinement of the partition: slots rowptr[b_r]..rowptr[b_{r+1}] = chunk r
    from anemoi_models_amd.distributed.shapes import split_bounds

    plan = runtime.build_edge_plan(z["bip.edge_index"].to(device), 70, 31)
    b = split_bounds(31, 4)
    for r in range(4):
        slots = plan.perm[int(plan.rowptr[b[r]]):int(plan.rowptr[b[r + 1]])].long().sort().values
        assert torch.equal(z["bip.edge_index"][:, slots.cpu()], z[f"bip.index{r}"])


def test_sort_edges_1hop_chunks_bit_exact(golden_index_ops):
    """The PRODUCT's edge partition (distributed/khop_edges.py) against the vectors recorded from the reference's
    ``sort_edges_1hop_chunks`` (reference distributed/khop_edges.py:88-130)."""
    _check_khop_against_golden(golden_index_ops, "cpu")


def test_sort_edges_1hop_sharding_identity_without_group(golden_index_ops):
    from anemoi_models_amd.distributed import khop_edges as K

    z = golden_index_ops
    a, i, sa, si = K.sort_edges_1hop_sharding(53, z["homo.edge_attr"], z["homo.edge_index"], None)
    assert a is z["homo.edge_attr"] and i is z["homo.edge_index"] and sa == [] and si == []


def test_get_k_hop_edges_two_hops():
    """directed k-hop: hop 2 adds the edges INTO the sources reached at hop 1 (PyG k_hop_subgraph contract)."""
    from anemoi_models_amd.distributed import khop_edges as K

    # chain 0 -> 1 -> 2 -> 3 plus a stray edge 4 -> 0
    ei = torch.tensor([[0, 1, 2, 4], [1, 2, 3, 0]])
    ea = torch.arange(4.0).view(4, 1)
    a1, i1 = K.get_k_hop_edges(torch.tensor([3]), ea, ei, 1)
    assert i1.tolist() == [[2], [3]] and a1.flatten().tolist() == [2.0]
    a2, i2 = K.get_k_hop_edges(torch.tensor([3]), ea, ei, 2)
    assert i2.tolist() == [[1, 2], [2, 3]] and a2.flatten().tolist() == [1.0, 2.0]


@pytest.mark.parametrize("world", [2, 3])
def test_shard_plan_processor_edges_are_the_reference_chunks(graph_o32, world):
    """Rank r of the node-partitioned forward owns exactly chunk r of ``sort_edges_1hop_chunks`` taken over the
    Morton-relabelled mesh (undoing the relabel gives back original edge ids, in original order)."""
    from anemoi_models_amd.distributed import khop_edges as K
    from anemoi_models_amd.distributed.partition import SimulatedRank
    from anemoi_models_amd.distributed.partition import build_shard_plan

    model = build_model(graph_o32)
    order, inv = model._mesh_order(torch.device("cpu"))
    ei = model.processor.edge_index_base
    n = order.shape[0]
    e_ids = torch.arange(ei.shape[1]).view(-1, 1)
    ids_l, idx_l = K.sort_edges_1hop_chunks(n, e_ids, inv[ei], world)
    for r in range(world):
        sp = build_shard_plan(model, SimulatedRank(r, world), torch.device("cpu"))
        mine = sp.proc.plan.perm.long().sort().values
        assert torch.equal(mine, ids_l[r].flatten())
        assert torch.equal(order[idx_l[r]], ei[:, mine])  # relabel undone: the original (src, dst) pairs
        assert int(idx_l[r][1].min()) >= sp.lo and int(idx_l[r][1].max()) < sp.hi


def test_graph_transformer_conv_forward_host_wiring(monkeypatch):
    """GraphTransformerConv.forward (reference layers/conv.py:98-142 call signature): plan, CSR permutation of the edge
    features and reshapes around the kernel, on the CPU stand-in, against oracle.gt_conv."""
    import _cpu_ops
    from anemoi_models_amd.layers.conv import GraphTransformerConv
    from oracle import reference_path as ref

    _cpu_ops.install(monkeypatch)
    g = torch.Generator().manual_seed(11)
    n_src, n_dst, e, h, d = 40, 25, 300, 4, 8
    ei = torch.stack([torch.randint(0, n_src, (e,), generator=g), torch.randint(0, n_dst - 1, (e,), generator=g)])
    q, k, v = (torch.randn(n, h, d, generator=g) for n in (n_dst, n_src, n_src))
    edges = torch.randn(e, h, d, generator=g)
    with torch.no_grad():
        got = GraphTransformerConv(out_channels=d).eval()(q, k, v, edges, ei, size=(n_src, n_dst))
    torch.testing.assert_close(got, ref.gt_conv(q, k, v, edges, ei, n_dst), atol=1e-5, rtol=1e-5)
    assert float(got[n_dst - 1].abs().max()) == 0.0  # isolated destination
    # dropout of the attention weights in training mode (layers/conv.py:140): the seed drawn from torch's generator, the mask
    # over CSR positions carried back to the caller's edge order, against the oracle with that mask
    conv = GraphTransformerConv(out_channels=d, dropout=0.35)
    torch.manual_seed(99)
    seed = int(torch.randint(0, 2**31 - 1, (1,)).item())
    torch.manual_seed(99)
    with torch.no_grad():
        dropped = conv(q, k, v, edges, ei)
        assert conv.dropout_args()[0] == 0.35 and conv.eval().dropout_args() == (0.0, 0, None)
    plan = conv._plans.get(ei, n_src, n_dst)
    keep = torch.empty(e, h, dtype=torch.float64)
    keep[plan.perm.long()] = _cpu_ops.edge_dropout_keep_mask(seed, 0.35, e, h)
    assert 0.55 < float(keep.mean()) < 0.75
    torch.testing.assert_close(dropped, ref.gt_conv(q, k, v, edges, ei, n_dst, dropout_p=0.35, keep=keep), atol=1e-5,
                               rtol=1e-5)


def test_reference_piecewise_api_of_blocks_and_mappers(graph_o32, monkeypatch):
    """The hooks the reference exposes next to ``forward`` (and its own tests call): block ``shard_qkve_heads`` /
    ``shard_output_seq`` (layers/block.py:366-414), mapper ``pre_process`` / ``post_process`` (layers/mapper.py:68-116,
    412-418, 690-694) -- shapes, shape bookkeeping and values for a single process."""
    import _cpu_ops
    from anemoi_models_amd.layers.block import GraphTransformerProcessorBlock
    from anemoi_models_amd.layers.mapper import GraphTransformerBackwardMapper, GraphTransformerForwardMapper

    _cpu_ops.install(monkeypatch)
    blk = GraphTransformerProcessorBlock(in_channels=64, hidden_dim=128, out_channels=64, edge_dim=5, num_heads=4)
    q, k, v, e = (torch.randn(30, 64) for _ in range(4))
    q3, k3, v3, e3 = blk.shard_qkve_heads(q, k, v, e, (10, 10, 10), 1)
    assert q3.shape == (30, 4, 16) and torch.equal(q3.reshape(30, 64), q) and torch.equal(e3.reshape(30, 64), e)
    assert torch.equal(blk.shard_output_seq(q3, (10, 10, 10), 1), q)

    sub = graph_o32[("data", "to", "hidden")]
    n_src, n_dst = graph_o32["data"].num_nodes, graph_o32["hidden"].num_nodes
    kw = dict(hidden_dim=64, trainable_size=4, num_heads=4, sub_graph=sub, sub_graph_edge_attributes=["edge_length", "edge_dirs"],
              src_grid_size=n_src, dst_grid_size=n_dst)
    fwd = GraphTransformerForwardMapper(in_channels_src=7, in_channels_dst=5, **kw).eval()
    x = (torch.randn(n_src, 7), torch.randn(n_dst, 5))
    shapes = ([[n_src, 7]], [[n_dst, 5]])
    with torch.no_grad():
        xs, xd, ss, sd = fwd.pre_process(x, shapes)
    assert xs.shape == (n_src, 64) and xd.shape == (n_dst, 64) and ss == [[n_src, 64]] and sd == [[n_dst, 64]]
    torch.testing.assert_close(xs, fwd.emb_nodes_src(x[0]), atol=1e-5, rtol=1e-5)
    assert fwd.post_process(xd, sd) is xd
    sub_b = graph_o32[("hidden", "to", "data")]
    bwd = GraphTransformerBackwardMapper(in_channels_src=64, in_channels_dst=5, hidden_dim=64, out_channels_dst=3,
                                         trainable_size=4, num_heads=4, sub_graph=sub_b,
                                         sub_graph_edge_attributes=["edge_length", "edge_dirs"], src_grid_size=n_dst,
                                         dst_grid_size=n_src).eval()
    with torch.no_grad():
        hs, hd, ss, sd = bwd.pre_process((torch.randn(n_dst, 64), torch.randn(n_src, 5)), ([[n_dst, 64]], [[n_src, 5]]))
        out = bwd.post_process(hd, sd)
    assert hs.shape == (n_dst, 64) and hd.shape == (n_src, 64) and sd == [[n_src, 64]] and out.shape == (n_src, 3)
    torch.testing.assert_close(out, bwd.node_data_extractor(hd), atol=1e-5, rtol=1e-5)


def test_mlp_leading_dimensions_and_unfused_activation(monkeypatch):
    """MLP on [B, N, F] inputs (nn.Linear semantics, reference tests/layers/test_mlp.py) and an activation the GEMM
    epilogue does not have (the reference takes any torch.nn activation by name): a torch op behind the Linear."""
    import _cpu_ops
    from anemoi_models_amd.layers.mlp import MLP

    _cpu_ops.install(monkeypatch)
    torch.manual_seed(3)
    for act in ("SiLU", "Tanh"):
        mlp = MLP(64, 128, 36, activation=act, layer_norm=True).eval()
        x = torch.randn(2, 50, 64)
        with torch.no_grad():
            y = mlp(x)
            want = mlp.model(x)  # the nn.Sequential itself on torch ops
        assert y.shape == (2, 50, 36)
        torch.testing.assert_close(y, want, atol=2e-5, rtol=2e-5)


def test_runs_of_destinations_sharing_three_sources():
    """runtime._runs3: the run lists of the decoder-style graphs (every destination exactly three in-edges) -- runs are
    maximal stretches of consecutive destinations with the same source SET, capped at 2; the packed permutation maps
    ascending source order back to CSR positions; any other graph gives None."""
    import torch

    from anemoi_models_amd import runtime

    g = torch.Generator().manual_seed(3)
    n, n_src = 3000, 500
    base = torch.randint(0, n_src - 3, (n,), generator=g)
    base = base[torch.arange(n) // 3 * 3]  # triples of destinations share a triangle ...
    base[7] = base[6] = base[5] = base[4] = base[3]  # ... one stretch of seven (3 .. 9 with the triple 9 .. 11 -> capped runs)
    base[8] = base[3]
    tri = torch.stack([base, base + 1, base + 2], 1)
    # every destination lists its three sources in its own order
    order = torch.stack([torch.randperm(3, generator=g) for _ in range(n)])
    src = torch.gather(tri, 1, order).reshape(-1)
    dst = torch.arange(n).repeat_interleave(3)
    shuffle = torch.randperm(3 * n, generator=g)
    plan = runtime.build_edge_plan(torch.stack([src[shuffle], dst[shuffle]]), n_src, n)
    run_ptr, perm = runtime._runs3(plan)
    assert run_ptr.dtype == torch.int32 and perm.dtype == torch.int32 and perm.shape == (run_ptr.shape[0] - 1,)
    assert run_ptr[0] == 0 and run_ptr[-1] == n and bool((run_ptr[1:] > run_ptr[:-1]).all())
    lens = (run_ptr[1:] - run_ptr[:-1])
    assert int(lens.max()) <= 2
    col = plan.col.view(n, 3).long()
    for r in range(run_ptr.shape[0] - 1):
        d0, d1 = int(run_ptr[r]), int(run_ptr[r + 1])
        sets = {tuple(sorted(col[d].tolist())) for d in range(d0, d1)}
        assert len(sets) == 1
        if d1 < n and d1 - d0 < 2:  # a run ends where the source set changes (or at the cap)
            assert tuple(sorted(col[d1].tolist())) not in sets
    for r in range(0, run_ptr.shape[0] - 1, 11):
        for k, d in enumerate(range(int(run_ptr[r]), int(run_ptr[r + 1]))):
            pos = [(int(perm[r]) >> (6 * k + 2 * s)) & 3 for s in range(3)]
            assert sorted(pos) == [0, 1, 2] and [int(col[d, q]) for q in pos] == sorted(col[d].tolist())
    # degree not uniformly three / duplicate sources: no runs
    assert runtime._runs3(runtime.build_edge_plan(torch.stack([src[:-3], dst[:-3]]), n_src, n)) is None
    dup = src.clone().view(n, 3)
    dup[5, 1] = dup[5, 0]
    assert runtime._runs3(runtime.build_edge_plan(torch.stack([dup.reshape(-1), dst]), n_src, n)) is None


def test_groups_of_destinations_sharing_three_sources():
    """runtime._groups3: every destination exactly once, sorted by source triple, a triple's destinations cut into groups of at
    most 8 wherever they lie in the destination order; the per-destination permutation maps ascending source order back to
    CSR positions; graphs that are not uniformly of degree three (or too short-grouped to pay) give None."""
    from anemoi_models_amd import runtime

    g = torch.Generator().manual_seed(5)
    n, n_src, n_tri = 4000, 700, 400
    base = torch.randint(0, n_src - 40, (n_tri,), generator=g)
    tri_of = torch.randint(0, n_tri, (n,), generator=g)  # ~10 destinations per triangle, scattered over the order
    tri_of[:30] = 7                                      # one triangle with > 8 + 8 destinations
    tri = torch.stack([base, base + 7, base + 31], 1)[tri_of]
    order = torch.stack([torch.randperm(3, generator=g) for _ in range(n)])
    src = torch.gather(tri, 1, order).reshape(-1)
    dst = torch.arange(n).repeat_interleave(3)
    shuffle = torch.randperm(3 * n, generator=g)
    plan = runtime.build_edge_plan(torch.stack([src[shuffle], dst[shuffle]]), n_src, n)
    grp_ptr, grp_perm, grp_dst = runtime._groups3(plan)
    assert grp_ptr.dtype == grp_perm.dtype == grp_dst.dtype == torch.int32
    assert grp_perm.shape == (n,) and torch.equal(torch.sort(grp_dst.long()).values, torch.arange(n))
    assert grp_ptr[0] == 0 and grp_ptr[-1] == n
    lens = grp_ptr[1:] - grp_ptr[:-1]
    assert int(lens.min()) >= 1 and int(lens.max()) <= 8 and n / lens.shape[0] > 4
    col = plan.col.view(n, 3).long()
    key = [tuple(sorted(col[d].tolist())) for d in range(n)]
    seen = []
    for r in range(grp_ptr.shape[0] - 1):
        ds = grp_dst[int(grp_ptr[r]):int(grp_ptr[r + 1])].tolist()
        assert len({key[d] for d in ds}) == 1 and ds == sorted(ds)
        seen.append(key[ds[0]])
    assert seen == sorted(seen)  # groups in ascending triple order; a triple's groups adjacent ...
    for a, b, la in zip(seen[:-1], seen[1:], lens[:-1].tolist()):
        assert a != b or la == 8  # ... and only a full group is followed by another one of the same triple
    for i in range(0, n, 13):
        d = int(grp_dst[i])
        pos = [(int(grp_perm[i]) >> (2 * s)) & 3 for s in range(3)]
        assert sorted(pos) == [0, 1, 2] and [int(col[d, q]) for q in pos] == sorted(col[d].tolist())
    assert runtime._groups3(runtime.build_edge_plan(torch.stack([src[:-3], dst[:-3]]), n_src, n)) is None
    i = torch.arange(n)
    lone = torch.stack([i % 690, 690 + i // 690, torch.full((n,), 699)], 1)  # every destination its own triple
    assert runtime._groups3(runtime.build_edge_plan(torch.stack([lone.reshape(-1), dst]), n_src, n)) is None


def test_edge_schedule_lists_cover_every_destination_once_and_balance_the_slots():
    """Host logic of the scheduled edge kernel (runtime.edge_schedule_lists): per XCD every destination of its range exactly
    once, at step i the i-th group of `slots` consecutive destinations (the L2 window of the round-robin kernel), lists end
    with >= 3 times -1 -- and on a multi-scale-mesh degree mix (6 ... 36) the busiest slot carries < 1.10 x the mean cost
    where the round-robin assignment carries > 1.15 x (the real ico-6 mesh in Morton order: 1.5 x)."""
    import torch

    from anemoi_models_amd.runtime import SCHED_U, SCHED_UNIT_COST, edge_schedule_lists

    g = torch.Generator().manual_seed(5)
    n, slots = 40962, 320
    steps = -(-((n + 7) // 8) // slots) + 3
    deg = torch.tensor([6, 12, 18, 24, 30, 36])[torch.multinomial(torch.tensor([.75, .1875, .047, .012, .003, .001]), n,
                                                                    replacement=True, generator=g)]
    sched = edge_schedule_lists(deg, slots, steps)
    assert sched.shape == (8, slots, steps) and sched.dtype == torch.int32
    cost = SCHED_UNIT_COST + torch.div(deg + SCHED_U - 1, SCHED_U, rounding_mode="floor").double()
    for x in range(8):
        n0, n1 = n * x // 8, n * (x + 1) // 8
        ids = sched[x]
        assert torch.equal(ids[ids >= 0].sort().values, torch.arange(n0, n1, dtype=torch.int32))
        assert bool((ids[:, -3:] == -1).all())
        for i in range(steps - 3):
            col = ids[:, i]
            col = col[col >= 0]
            lo = n0 + i * slots
            assert col.numel() == min(slots, max(n1 - lo, 0)) and (col.numel() == 0 or (col.min() >= lo and col.max() < lo + slots))
        filled = (ids >= 0)
        assert bool((filled[:, 1:] <= filled[:, :-1]).all())  # no hole inside a list
        load = torch.where(filled, cost[ids.clamp_min(0).long()], torch.zeros(())).sum(1)
        rr = torch.stack([cost[n0 + s:n1:slots].sum() for s in range(slots)])
        assert float(load.max() / load.mean()) < 1.10 < 1.15 < float(rr.max() / rr.mean()), (load.max() / load.mean(), rr.max() / rr.mean())
    uniform = edge_schedule_lists(torch.full((1000,), 3), 125, 4)
    assert torch.equal(uniform[0, :, 0], torch.arange(125, dtype=torch.int32)) and int((uniform >= 0).sum()) == 1000


def test_edge_tile_lists_cover_every_destination_once_within_their_caps():
    """``runtime.edge_tile_lists`` (host lists of the LDS-tile edge kernel, round 6): every destination in exactly one tile of
    its own XCD's range, tiles within the caps, ``slot -> source`` reproduces the CSR's source column, the packed
    ``(first edge << 8 | degree)`` words match the row pointers, passes ordered by descending in-degree; a destination that
    alone exceeds a cap makes the graph ineligible (``None``)."""
    g = torch.Generator().manual_seed(3)
    n_dst, n_src = 1000, 900
    deg = torch.randint(0, 14, (n_dst,), generator=g)
    deg[17] = 40
    dst = torch.repeat_interleave(torch.arange(n_dst), deg)
    src = (dst * n_src // n_dst + torch.randint(-20, 21, dst.shape, generator=g)).clamp_(0, n_src - 1)
    plan = runtime.build_edge_plan(torch.stack([src, dst]), n_src, n_dst)
    t = runtime.edge_tile_lists(plan.rowptr, plan.col, src_cap=48, edge_cap=160)
    assert t is not None and t.hdr.shape == (t.n_tiles, 8) and t.dst.shape == (t.n_tiles, 32, 2) and t.xcd.shape == (9,)
    rp, cl = plan.rowptr.long(), plan.col.long()
    seen = torch.zeros(n_dst, dtype=torch.int64)
    for x in range(8):
        for ti in range(int(t.xcd[x]), int(t.xcd[x + 1])):
            e0, ne, so, ns, slo, nd = t.hdr[ti, :6].tolist()
            assert 1 <= nd <= 32 and 1 <= ns <= 48 and ne <= 160 and slo % 16 == 0
            assert torch.equal(t.src[so:so + ns].long()[t.slot[slo:slo + ne].long()], cl[e0:e0 + ne])
            degs = []
            for node, pk in t.dst[ti].tolist():
                if node < 0:
                    continue
                assert n_dst * x // 8 <= node < n_dst * (x + 1) // 8  # the tile belongs to its XCD's destination range
                assert int(rp[node]) == e0 + (pk >> 8) and int(rp[node + 1] - rp[node]) == (pk & 255)
                seen[node] += 1
                degs.append(pk & 255)
            assert len(degs) == nd and degs == sorted(degs, reverse=True)
    assert bool((seen == 1).all())
    assert int(t.hdr[:, 1].sum()) == plan.num_edges > 1.5 * t.src.shape[0]  # neighbours share sources: re-use inside a tile
    assert runtime.edge_tile_lists(plan.rowptr, plan.col, src_cap=12, edge_cap=160) is None  # destination 17: 40 edges, > 12 sources ...
    assert runtime.edge_tile_lists(plan.rowptr, plan.col, src_cap=48, edge_cap=32) is None   # ... alone beyond either cap


def test_grad_sink_hands_out_the_stacked_gradient_without_a_copy():
    """``autograd.GradSink`` / ``_Unstack`` (training route of a processor): when the gradient of block i's weight IS slot i of
    the sink, the gradient of the stacked weight is the sink's buffer itself (no ``torch.stack``); anything else -- a missing
    gradient, a tensor that lives elsewhere -- falls back to stacking, with the same values."""
    from anemoi_models_amd import autograd

    count, n, k = 3, 4, 5

    class _FromSlot(torch.autograd.Function):  # stands for _Linear: its weight gradient is written into the sink's slot
        @staticmethod
        def forward(ctx, w, b, sink, i, elsewhere):
            ctx.sink, ctx.i, ctx.elsewhere = sink, i, elsewhere
            return (w.sum() + b.sum()).reshape(1)

        @staticmethod
        def backward(ctx, g):
            slot = ctx.sink.slot(ctx.i)
            slot[: n * k] = float(ctx.i + 1)
            slot[n * k:] = float(-ctx.i - 1)
            dw, db = slot[: n * k].view(n, k), slot[n * k:]
            if ctx.elsewhere:
                dw = dw.clone()
            return dw, db, None, None, None

    for elsewhere in (False, True):
        w_leaf = torch.randn(count, n, k, requires_grad=True)
        b_leaf = torch.randn(count, n, requires_grad=True)
        w, b = w_leaf * 1.0, b_leaf * 1.0  # (the stacked operands are results of the fold algebra, not leaves)
        seen = {}
        w.register_hook(lambda g: seen.__setitem__("w", g))
        b.register_hook(lambda g: seen.__setitem__("b", g))
        sink = autograd.GradSink(count, n, k, True, "cpu", stacked_parts=2)
        ws = autograd._Unstack.apply(w, sink, "w")
        bs = autograd._Unstack.apply(b, sink, "b")
        total = sum(_FromSlot.apply(ws[i], bs[i], sink, i, elsewhere and i == 1) for i in range(count))
        buf_ptr = []
        real_slot = sink.slot

        def slot(i, _real=real_slot):
            t = _real(i)
            buf_ptr.append(sink.buf.data_ptr())
            return t

        sink.slot = slot
        total.sum().backward()
        want_w = torch.arange(1.0, count + 1)[:, None, None].expand(count, n, k)
        want_b = -torch.arange(1.0, count + 1)[:, None].expand(count, n)
        assert torch.equal(w_leaf.grad, want_w) and torch.equal(b_leaf.grad, want_b)
        assert sink.buf is None or elsewhere  # handed over (both parts collected) -- or kept: the "w" part was stacked instead
        assert seen["b"].data_ptr() == buf_ptr[0] + n * k * 4
        if not elsewhere:  # zero-copy: the stacked gradient IS the sink's buffer
            assert seen["w"].data_ptr() == buf_ptr[0] and seen["w"].stride() == (n * k + n, k, 1)
        else:
            assert seen["w"].data_ptr() != buf_ptr[0]
    # a block without a gradient: zeros for it, the others stacked
    w = torch.randn(count, n, k, requires_grad=True)
    ws = autograd._Unstack.apply(w, None, "w")
    (ws[0].sum() * 2.0 + ws[2].sum() * 3.0).backward()
    assert torch.equal(w.grad[0], torch.full((n, k), 2.0)) and torch.equal(w.grad[1], torch.zeros(n, k))
    assert torch.equal(w.grad[2], torch.full((n, k), 3.0))


def test_stacked_fold_of_a_processor_equals_the_per_block_fold():
    """``autograd._gt_stacked_fold`` (the lin_edge fold and the ``x_r|q|k|v|u`` / ``projection|t`` weight assembly of ALL blocks
    of a processor from three batched einsums, training route) against the per-block algebra it replaces
    (``autograd._lin_edge_fold`` + the concatenations of ``gt_processor_block``): same matrices, and the same gradients for
    every parameter under a random cotangent."""
    from anemoi_models_amd import autograd

    c, h, edge_dim, up, count = 64, 8, 3, 4, 3
    g = torch.Generator().manual_seed(9)

    def block():
        names = {"lin_self": (c, c), "lin_query": (c, c), "lin_key": (c, c), "lin_value": (c, c), "lin_edge": (c, edge_dim),
                 "projection": (c, c)}
        sd = {}
        for n, shape in names.items():
            sd[f"b.{n}.weight"] = torch.randn(*shape, generator=g).requires_grad_()
            sd[f"b.{n}.bias"] = torch.randn(shape[0], generator=g).requires_grad_()
        return sd

    sds = [block() for _ in range(count)]
    w_in, b_in, w_p = autograd._gt_stacked_fold(sds, "b", c, h, up, "cpu")
    assert w_in.shape == (count, 4 * c + h * up, c) and b_in.shape == (count, 4 * c + h * up) and w_p.shape == (count, c, c + h * up)
    cot = [torch.randn(t.shape, generator=g) for t in (w_in, b_in, w_p)]
    (w_in * cot[0]).sum().add((b_in * cot[1]).sum()).add((w_p * cot[2]).sum()).backward()
    got = [{k: v.grad.clone() for k, v in sd.items() if v.grad is not None} for sd in sds]
    for sd in sds:
        for v in sd.values():
            v.grad = None
    for i, sd in enumerate(sds):
        w_u, b_u, w_t = autograd._lin_edge_fold(sd, "b", c, h, up, "cpu")
        gp = lambda n: sd["b." + n]  # noqa: E731
        wi = torch.cat([gp("lin_self.weight"), gp("lin_query.weight"), gp("lin_key.weight"), gp("lin_value.weight"), w_u], 0)
        bi = torch.cat([gp("lin_self.bias"), gp("lin_query.bias"), gp("lin_key.bias"), gp("lin_value.bias"), b_u], 0)
        wp = torch.cat([gp("projection.weight"), w_t], 1)
        assert torch.allclose(w_in[i], wi, rtol=1e-6, atol=1e-6) and torch.allclose(b_in[i], bi, rtol=1e-6, atol=1e-6)
        assert torch.allclose(w_p[i], wp, rtol=1e-6, atol=1e-6)
        (wi * cot[0][i]).sum().add((bi * cot[1][i]).sum()).add((wp * cot[2][i]).sum()).backward()
        for k, v in sd.items():
            if k == "b.projection.bias":  # (never enters the fold: the GEMM takes it as it is)
                assert v.grad is None and k not in got[i]
                continue
            assert torch.allclose(got[i][k], v.grad, rtol=1e-5, atol=1e-5), (i, k)


# ---------------------------------------------------------------------------------------------
# The listings of the inline-asm kernels: nothing touches an MFMA result before its wait states have passed
# ---------------------------------------------------------------------------------------------
def _audit_module():
    import importlib.util

    spec = importlib.util.spec_from_file_location(
        "isa_hazard_audit", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "isa_hazard_audit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_isa_hazard_audit_reports_a_read_right_behind_its_mfma(tmp_path):
    """The auditor itself, on a hand-written listing: the round-5 bug pattern (a v_max on a score register directly behind the
    MFMA writing it), a wait that is long enough, an accumulate chain (exempt) and a short wait."""
    audit = _audit_module()
    listing = tmp_path / "k.s"
    listing.write_text("\n".join([
        "_Z1kv:",                                                               # line 1
        "\t;;#ASMSTART",
        "\tv_mfma_f32_32x32x16_bf16 v[16:31], v[56:59], a[0:3], v[16:31]",   # 3: chain link: exempt ...
        "\t;;#ASMEND",
        "\t;;#ASMSTART",
        "\tv_mfma_f32_32x32x16_bf16 v[16:31], v[60:63], a[4:7], v[16:31]",   # 6: ... and this one is read too early
        "\t;;#ASMEND",
        "\tv_max_f32_e32 v16, v16, v16",                                      # 8
        "\t;;#ASMSTART",
        "\tv_mfma_f32_32x32x16_bf16 v[0:15], v[56:59], a[0:3], 0",           # 10
        "\t;;#ASMEND",
        "\ts_nop 15",
        "\tv_max_f32_e32 v0, v0, v0",                                         # 13: behind 16 states: fine
        "\t;;#ASMSTART",
        "\tv_mfma_f32_16x16x32_bf16 a[0:3], v[56:59], v[60:63], a[0:3]",     # 15
        "\t;;#ASMEND",
        "\ts_nop 3",
        "\tv_accvgpr_read_b32 v1, a2",                                        # 18: 4 states of 8: too early
        "\ts_nop 15",
        ".LBB0_1:",                                                             # a loop whose last product is read at its top
        "\tv_add_f32_e32 v2, v40, v40",                                       # 21: read of v[32:47] through the back edge
        "\ts_nop 15",
        "\t;;#ASMSTART",
        "\tv_mfma_f32_32x32x16_bf16 v[32:47], v[56:59], a[0:3], 0",          # 24
        "\t;;#ASMEND",
        "\ts_cbranch_scc1 .LBB0_1",
        "\ts_nop 15",
        "\tv_mfma_f32_32x32x16_bf16 v[48:63], v[56:59], v[60:63], v[48:63]",  # 28: compiler-generated on both sides:
        "\tv_max_f32_e32 v3, v48, v48",                                       # 29: reported only with everything=True
        "\ts_endpgm",
    ]) + "\n")
    found = audit.audit(str(listing))
    assert [(f[1], f[3]) for f in found] == [(6, 8), (15, 18), (24, 21)], found
    assert (28, 29) in [(f[1], f[3]) for f in audit.audit(str(listing), everything=True)]
    # second rule: a vector write needs 2 wait states before an MFMA reads it as an operand
    listing.write_text("\n".join([
        "_Z1kv:",
        "\tv_mov_b32_e32 v56, 0",
        "\tv_mfma_f32_32x32x16_bf16 a[0:15], v[56:59], v[60:63], a[0:15]",   # right behind the write: reported
        "\tv_mov_b32_e32 v60, 0",
        "\ts_nop 1",
        "\tv_mfma_f32_32x32x16_bf16 a[16:31], v[56:59], v[60:63], a[16:31]",  # behind 2 states: fine
        "\tv_accvgpr_write_b32 a3, v1",
        "\ts_add_i32 s0, s0, 1",
        "\tv_mfma_f32_32x32x16_bf16 a[0:15], v[56:59], v[60:63], a[0:15]",   # one state only: reported
        "\ts_endpgm",
    ]) + "\n")
    found = audit.audit_operands(str(listing))
    assert [(f[1], f[3]) for f in found] == [(3, 2), (9, 7)], found
    # third rule: the data registers of a wide buffer store with an SGPR soffset stay untouched for 2 wait states
    listing.write_text("\n".join([
        "_Z1kv:",
        "\tbuffer_store_dwordx4 v[106:109], v104, s[0:3], s93 offen",
        "\tv_mov_b32_e32 v107, 0",                                             # the very next instruction: reported
        "\tbuffer_store_dwordx4 v[110:113], v104, s[0:3], s93 offen",
        "\ts_nop 1",
        "\tv_mov_b32_e32 v110, 0",                                             # behind 2 states: fine
        "\tbuffer_store_dwordx4 v[114:117], v104, s[0:3], 0 offen",
        "\tv_mov_b32_e32 v114, 0",                                             # no SGPR soffset: hipcc's own padding applies
        "\tbuffer_store_dwordx2 v[118:119], v104, s[0:3], s93 offen",
        "\tv_mov_b32_e32 v118, 0",                                             # 8 bytes: no hazard
        "\ts_endpgm",
    ]) + "\n")
    found = audit.audit_stores(str(listing))
    assert [(f[1], f[3]) for f in found] == [(2, 3)], found


def test_no_mfma_result_is_touched_before_its_wait_states_in_the_built_kernels():
    """Every listing the build kept (anemoi_models_amd/_build.py::ASM_SOURCES): zero early touches.  Skipped only where the
    library was not built from source in this tree (no listing to read)."""
    from anemoi_models_amd import _build

    audit = _audit_module()
    seen = 0
    for name in _build.ASM_SOURCES:
        path = _build.device_listing(name)
        if path is None:
            continue
        seen += 1
        found = audit.audit(path)
        assert not found, f"{name}.hip: {len(found)} early touches of an MFMA destination, first: {found[0]}"
        found = audit.audit_operands(path)
        assert not found, f"{name}.hip: {len(found)} MFMA operands written fewer than 2 wait states ahead, first: {found[0]}"
        found = audit.audit_stores(path)
        assert not found, f"{name}.hip: {len(found)} wide buffer stores whose data is overwritten too early, first: {found[0]}"
    if seen == 0:
        pytest.skip("no device listing in anemoi_models_amd/lib/obj (run __graft_entry__.build() from source)")


def test_folded_edge_route_and_conv_head_size():
    """Which edge route the differentiable GraphTransformer blocks take per shape (``autograd.folded_edge_route``) and the head
    size the explicit-edge conv kernels run a head at (``autograd.conv_head_size``): pure host logic behind the shapes the
    round-6 sweeps found refused."""
    import torch

    from anemoi_models_amd import autograd

    f32, bf16 = torch.float32, torch.bfloat16
    # the benchmark shapes stay on the folded kernels
    assert autograd.folded_edge_route(bf16, 1024, 16, 12) and autograd.folded_edge_route(bf16, 512, 16, 12)
    assert autograd.folded_edge_route(f32, 64, 16, 4) and autograd.folded_edge_route(bf16, 64, 16, 4)  # D = 4: f32 edge phase
    assert not autograd.folded_edge_route(bf16, 128, 8, 40)        # more edge attributes than the folded kernels carry
    assert not autograd.folded_edge_route(f32, 96, 8, 12)          # heads of 12: three f32 lanes
    assert not autograd.folded_edge_route(bf16, 192, 16, 12)       # heads of 12 in bf16: f32 edge phase, still three lanes
    assert not autograd.folded_edge_route(bf16, 192, 4, 4)         # heads of 48: six bf16 lanes
    assert not autograd.folded_edge_route(f32, 160, 16, 12)        # heads of 10: not a multiple of 4
    assert not autograd.folded_edge_route(bf16, 64, 1, 4)          # one bf16 head: H * up = 4 breaks the 16-byte row pitch
    assert autograd.folded_edge_route(bf16, 64, 1, 8) and autograd.folded_edge_route(f32, 64, 1, 4)
    assert not autograd.folded_edge_route(f32, 64, 3, 4)           # channels that do not split into the heads
    for d, dtype, want in ((4, f32, 4), (5, f32, 8), (12, f32, 16), (20, f32, 32), (48, f32, 64), (64, f32, 64), (1, f32, 4),
                           (8, bf16, 8), (5, bf16, 8), (12, bf16, 16), (48, bf16, 64), (96, bf16, 128), (128, bf16, 128)):
        assert autograd.conv_head_size(d, dtype) == want, (d, dtype)
    for d, dtype in ((65, f32), (129, bf16)):
        with pytest.raises(NotImplementedError):
            autograd.conv_head_size(d, dtype)

