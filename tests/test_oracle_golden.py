"""The CPU oracle must reproduce the vectors recorded from the real reference (tests/golden/make_golden.py)."""

import torch

from conftest import split_prefix
from oracle import reference_path as ref

ATOL, RTOL = 2e-5, 2e-5  # fp32 CPU vs fp32 CPU: only summation-order noise is allowed


def graph_tensors(g):
    def cat(key):
        s = g[key]
        return torch.cat([s["edge_length"], s["edge_dirs"]], dim=1)

    return {
        "enc_edge_index": g[("data", "to", "hidden")].edge_index, "enc_edge_attr": cat(("data", "to", "hidden")),
        "proc_edge_index": g[("hidden", "to", "hidden")].edge_index, "proc_edge_attr": cat(("hidden", "to", "hidden")),
        "dec_edge_index": g[("hidden", "to", "data")].edge_index, "dec_edge_attr": cat(("hidden", "to", "data")),
    }


def _strip(sd):
    return {("x." + k): v for k, v in sd.items()}


def test_gt_processor_block(golden_blocks):
    b = golden_blocks
    sd = _strip(split_prefix(b, "gtp.sd."))
    y = ref.gt_processor_block(sd, "x", b["gtp.x"], b["gtp.edge_attr"], b["gtp.edge_index"], 16)
    torch.testing.assert_close(y, b["gtp.y"], atol=ATOL, rtol=RTOL)


def test_gt_mapper_block_and_chunk_invariance(golden_blocks):
    b = golden_blocks
    sd = _strip(split_prefix(b, "gtm.sd."))
    args = (sd, "x", b["gtm.x_src"], b["gtm.x_dst"], b["gtm.edge_attr"], b["gtm.edge_index"], 16)
    y = ref.gt_mapper_block(*args)
    torch.testing.assert_close(y, b["gtm.y_dst"], atol=ATOL, rtol=RTOL)
    # reference test_GraphTransformerMapperBlock_chunking (tests/layers/block/test_block_graphtransformer.py:339-377)
    for chunks in (2, 3, 7):
        yc = ref.gt_mapper_block(*args, num_chunks=chunks)
        assert torch.allclose(y, yc, atol=1e-4)


def test_gnn_block(golden_blocks):
    b = golden_blocks
    sd = _strip(split_prefix(b, "gnn.sd."))
    y, e_new = ref.gnn_processor_block(sd, "x", b["gnn.x"], b["gnn.edge_attr"], b["gnn.edge_index"])
    torch.testing.assert_close(y, b["gnn.y"], atol=ATOL, rtol=RTOL)
    torch.testing.assert_close(e_new, b["gnn.edges_new"], atol=ATOL, rtol=RTOL)


def test_transformer_block(golden_blocks):
    b = golden_blocks
    sd = _strip(split_prefix(b, "tfm.sd."))
    att = ref.mhsa(sd, "x.attention", b["tfm.x"], 2, 8)
    torch.testing.assert_close(att, b["tfm.att"], atol=ATOL, rtol=RTOL)
    y = ref.transformer_block(sd, "x", b["tfm.x"], 2, 8)
    torch.testing.assert_close(y, b["tfm.y"], atol=ATOL, rtol=RTOL)


def test_index_ops_bit_exact(golden_index_ops):
    z = golden_index_ops
    attr_l, idx_l = ref.sort_edges_1hop_chunks(53, z["homo.edge_attr"], z["homo.edge_index"], 5)
    for i in range(5):
        assert torch.equal(idx_l[i], z[f"homo.index{i}"]) and torch.equal(attr_l[i], z[f"homo.attr{i}"])
    attr_l, idx_l = ref.sort_edges_1hop_chunks((70, 31), z["bip.edge_attr"], z["bip.edge_index"], 4)
    for i in range(4):
        assert torch.equal(idx_l[i], z[f"bip.index{i}"]) and torch.equal(attr_l[i], z[f"bip.attr{i}"])
    inc = torch.tensor([[70], [31]], dtype=torch.int64)
    assert torch.equal(ref.expand_edges(z["expand.edge_index"], inc, 3), z["expand.out"])


def _model(gold, graph, processor, mappers="GraphTransformer"):
    sd = split_prefix(gold, "sd.")
    y, st = ref.model_forward(
        sd, graph_tensors(graph), gold["x"], num_heads=16, num_layers=4, num_chunks=2,
        prognostic_in=range(10), prognostic_out=range(10), processor=processor, return_stages=True, mappers=mappers,
    )
    torch.testing.assert_close(st["x_latent"], gold["stage.encoder"], atol=ATOL, rtol=RTOL)
    torch.testing.assert_close(st["x_proc"], gold["stage.processor"], atol=5 * ATOL, rtol=5 * RTOL)
    torch.testing.assert_close(y, gold["y"], atol=5 * ATOL, rtol=5 * RTOL)


def test_model_gt(golden_cfg1_gt, graph_o32):
    _model(golden_cfg1_gt, graph_o32, "GraphTransformer")


def test_model_gt_block_stages(golden_cfg1_gt, graph_o32):
    gold = golden_cfg1_gt
    sd = split_prefix(gold, "sd.")
    g = graph_tensors(graph_o32)
    _, outs = ref.gt_processor(sd, "processor", gold["stage.encoder"], g["proc_edge_attr"], g["proc_edge_index"], 1, 4,
                               2, 16, return_all=True)
    for i, o in enumerate(outs):
        torch.testing.assert_close(o, gold[f"stage.block{i}"], atol=5 * ATOL, rtol=5 * RTOL)


def test_model_gnn(golden_cfg1_gnn, graph_o32):
    _model(golden_cfg1_gnn, graph_o32, "GNN")


def test_model_gnn_mappers(golden_cfg1_gnn_all, graph_o32):
    _model(golden_cfg1_gnn_all, graph_o32, "GNN", mappers="GNN")


def test_model_transformer(golden_cfg1_tfm, graph_o32):
    _model(golden_cfg1_tfm, graph_o32, "Transformer")


def test_gt_conv_matches_dense_formulation():
    """Independent O(N_dst x E) dense check of the PyG-semantics restatement on a tiny graph."""
    g = torch.Generator().manual_seed(3)
    ns, nd, e, h, d = 7, 5, 23, 2, 3
    q, k, v = torch.randn(nd, h, d, generator=g), torch.randn(ns, h, d, generator=g), torch.randn(ns, h, d, generator=g)
    ea = torch.randn(e, h, d, generator=g)
    ei = torch.stack([torch.randint(0, ns, (e,), generator=g), torch.randint(0, nd - 1, (e,), generator=g)])
    out = ref.gt_conv(q, k, v, ea, ei, nd)
    dense = torch.zeros(nd, h, d, dtype=torch.float64)
    for i in range(nd):
        edges = [n for n in range(e) if int(ei[1, n]) == i]
        if not edges:
            continue
        for hh in range(h):
            s = torch.tensor([float((q[i, hh].double() * (k[ei[0, n], hh] + ea[n, hh]).double()).sum()) / d**0.5
                              for n in edges], dtype=torch.float64)
            w = torch.softmax(s, 0)
            for wn, n in zip(w, edges):
                dense[i, hh] += wn * (v[ei[0, n], hh] + ea[n, hh]).double()
    torch.testing.assert_close(out.double(), dense, atol=1e-5, rtol=1e-5)
    assert torch.all(out[nd - 1] == 0)  # isolated destination stays exactly zero


def hier_graph_tensors(g, hidden=("hidden_1", "hidden_2"), data="data"):
    def put(out, prefix, key):
        st = g[key]
        out[prefix + ".edge_index"] = st.edge_index
        out[prefix + ".edge_attr"] = torch.cat([st["edge_length"], st["edge_dirs"]], dim=1)

    out = {}
    put(out, "encoder", (data, "to", hidden[0]))
    put(out, "decoder", (hidden[0], "to", data))
    for h in hidden:
        put(out, f"down_level_processor.{h}", (h, "to", h))
        put(out, f"up_level_processor.{h}", (h, "to", h))
    for fine, coarse in zip(hidden[:-1], hidden[1:]):
        put(out, f"downscale.{fine}", (fine, "to", coarse))
        put(out, f"upscale.{coarse}", (coarse, "to", fine))
    return out


def test_hierarchical_model(golden_hier_gt, graph_hier):
    """oracle.hierarchical_forward == the real AnemoiModelEncProcDecHierarchical (reference models/hierarchical.py)."""
    gold = golden_hier_gt
    sd = split_prefix(gold, "sd.")
    y = ref.hierarchical_forward(sd, hier_graph_tensors(graph_hier), gold["x"], hidden=["hidden_1", "hidden_2"],
                                 num_heads=16, level_layers=2, prognostic_in=list(range(10)),
                                 prognostic_out=list(range(10)))
    torch.testing.assert_close(y, gold["y"], atol=ATOL, rtol=RTOL)


NORMALIZER_METHODS = {"default": "mean-std", "min-max": ["prog_3"], "max": ["prog_4"], "std": ["prog_5"],
                      "none": ["forc_0"], "remap": {"prog_7": "prog_6"}}  # as in tests/golden/make_golden.py


def test_normalizer_and_predict_step(golden_interface, graph_o32):
    """oracle.normalizer_affine / predict_step == the real InputNormalizer + AnemoiModelInterface.predict_step
    (reference preprocessing/normalizer.py, interface/__init__.py) built on the real IndexCollection."""
    gold = golden_interface
    sd = split_prefix(gold, "sd.")
    names = [f"prog_{i}" for i in range(10)] + [f"forc_{i}" for i in range(2)] + ["diag_0"]
    stats = {k: v.numpy() for k, v in split_prefix(gold, "stat.").items()}
    mul, add = ref.normalizer_affine(NORMALIZER_METHODS, {n: i for i, n in enumerate(names)}, stats)
    torch.testing.assert_close(mul, sd["pre_processors.processors.normalizer._norm_mul"], atol=0, rtol=1e-6)
    torch.testing.assert_close(add, sd["pre_processors.processors.normalizer._norm_add"], atol=1e-6, rtol=1e-6)
    y = ref.predict_step(sd, graph_tensors(graph_o32), gold["batch"], multi_step=2, num_heads=16, num_layers=4,
                         num_chunks=2, prognostic_in=list(range(10)), prognostic_out=list(range(10)))
    torch.testing.assert_close(y, gold["y"], atol=1e-4, rtol=1e-4)


def test_rollout_matches_reference_steps(golden_interface, graph_o32):
    """oracle.rollout == three applications of the real reference interface (model + InputNormalizer) chained by the
    caller's advance_input loop as recorded by tests/golden/make_golden.py::golden_interface."""
    gold = golden_interface
    sd = split_prefix(gold, "sd.")
    y = ref.rollout(sd, graph_tensors(graph_o32), gold["batch"], 3, gold["rollout_forcings"], multi_step=2,
                    num_heads=16, num_layers=4, num_chunks=2, prognostic_in=list(range(10)),
                    prognostic_out=list(range(10)), forcing_in=[10, 11])
    assert y.shape == gold["rollout_y"].shape
    torch.testing.assert_close(y[0], gold["y"], atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(y, gold["rollout_y"], atol=5e-4, rtol=5e-4)
