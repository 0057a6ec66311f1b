"""Embedding -> LayerNorm -> Linear fold of the GraphTransformer mappers (layers/mapper.py::_embedded,
runtime.fold_embedded_layer_norm / embedding_stats_operator): the q / k / v GEMMs of a mapper block run on the raw
node features [x | 1 | 0-pad] (K = padded feature count) instead of on the embedded rows (K = hidden width).

Reference chain that is restated: layers/mapper.py:322-331 (emb_nodes_src / emb_nodes_dst), layers/block.py:516-528
(layer_norm1 / layer_norm2, lin_key / lin_value / lin_self / lin_query)."""

import os

import pytest
import torch

import _cpu_ops
from oracle import reference_path as ref


def _build(graph, channels, layers):
    from anemoi_models_amd.models import AnemoiModelEncProcDec
    from anemoi_models_amd.utils.indices import SimpleDataIndices
    from anemoi_models_amd.utils.presets import model_config

    idx = SimpleDataIndices(n_prognostic=10, n_forcing=2, n_diagnostic=1)
    torch.manual_seed(4321)
    model = AnemoiModelEncProcDec(model_config=model_config("GraphTransformer", channels, layers, 16), data_indices=idx,
                                  graph_data=graph)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("trainable"):
                p.normal_(0.0, 0.1)
            elif "emb_nodes" in name and name.endswith("bias"):
                p.normal_(0.3, 0.5)  # a channel mean the LayerNorm has to remove
    return model.eval(), idx


def _oracle(model, graph, x, layers):
    from test_oracle_golden import graph_tensors

    sd = {k: v.clone() for k, v in model.state_dict().items()}
    return ref.model_forward(sd, graph_tensors(graph), x, num_heads=16, num_layers=layers, num_chunks=2,
                             prognostic_in=range(10), prognostic_out=range(10))


def rel_err(a, b):
    return float((a.float().cpu() - b.float().cpu()).abs().max() / b.float().abs().max())


@pytest.mark.parametrize("c,k_in,n,rows,with_bias", [(1024, 192, 96, 300, True), (128, 36, 40, 50, True),
                                                      (64, 12, 24, 20, False), (32, 40, 16, 30, True)])
def test_fold_algebra_f64(c, k_in, n, rows, with_bias):
    """rstd * (F x_aug) + b' == Linear(LayerNorm(emb(x))), and the side product T x_aug has per row mean 0 and
    mean(y^2) == var(emb(x)) -- any rank (the last case has more features than channels)."""
    from anemoi_models_amd import runtime

    g = torch.Generator().manual_seed(c + k_in)
    dd = dict(dtype=torch.float64, generator=g)
    e, be = 0.2 * torch.randn(c, k_in, **dd), (0.5 + torch.randn(c, **dd)) if with_bias else None
    w, b = 0.1 * torch.randn(n, c, **dd), torch.randn(n, **dd)
    gamma, beta = 0.5 + torch.rand(c, **dd), 0.1 * torch.randn(c, **dd)
    x = torch.randn(rows, k_in, **dd)
    kp, one = (k_in + 1 + 63) // 64 * 64, k_in
    xa = torch.zeros(rows, kp, dtype=torch.float64)
    xa[:, :k_in], xa[:, one] = x, 1.0
    h = x @ e.T + (0 if be is None else be)
    want = torch.nn.functional.layer_norm(h, (c,), gamma, beta, 1e-5) @ w.T + b

    f, bp, zero = runtime.fold_embedded_layer_norm(w, b, gamma, beta, e, be, kp, one, torch.float64)
    t = runtime.embedding_stats_operator(e, be, kp, one, torch.float64)
    assert t.shape[0] % 256 == 0 and t.shape[1] == kp and float(zero.abs().max()) == 0.0
    y = xa @ t.T
    var = h.var(dim=1, unbiased=False)
    assert float(y.mean(dim=1).abs().max()) < 1e-12
    torch.testing.assert_close((y * y).mean(dim=1), var, rtol=1e-10, atol=1e-12)
    got = torch.rsqrt(var + 1e-5)[:, None] * (xa @ f.T) + bp.double()
    torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-6)  # b' is returned in f32


def test_mapper_fold_host_wiring(graph_o32, monkeypatch):
    """The launch sequence with the fold (bf16 route on the CPU stand-in kernels) computes the reference function, with
    and without the fold, and with it the source embedding of the encoder is never formed."""
    from anemoi_models_amd import ops

    _cpu_ops.install(monkeypatch)
    monkeypatch.setenv("ANEMOI_AMD_DTYPE", "bf16")
    model, idx = _build(graph_o32, 256, 2)
    x = torch.randn(1, 2, 1, graph_o32["data"].num_nodes, idx.num_input, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        want = _oracle(model, graph_o32, x, 2)
    shapes = []
    real = ops.linear

    def spy(xx, w, *a, **kw):
        shapes.append((xx.shape[0], w.shape[0], w.shape[1]))
        return real(xx, w, *a, **kw)

    monkeypatch.setattr(ops, "linear", spy)
    n_grid = graph_o32["data"].num_nodes
    with torch.no_grad():
        monkeypatch.setenv("ANEMOI_AMD_EMBED_FOLD", "1")
        folded = model(x)
        with_fold = list(shapes)
        shapes.clear()
        monkeypatch.setenv("ANEMOI_AMD_EMBED_FOLD", "0")
        plain = model(x)
    assert rel_err(plain, want) < 3e-2 and rel_err(folded, want) < 3e-2
    # without the fold the grid rows pass a K = 256 GEMM (k | v of the encoder, x_r | q | u of the decoder); with it
    # every GEMM that starts from the embedding reads the 128 padded feature columns (37 features + the constant 1)
    assert (n_grid, 512, 256) in shapes and (n_grid, 512, 256) not in with_fold
    assert (n_grid, 512, 128) in with_fold
    # encoder source embedding (grid rows, 256 outputs from the features) is not formed: only the decoder's remains
    assert sum(1 for s in with_fold if s == (n_grid, 256, 128)) == 1 + 1  # + the [rows, 256] statistics side product
    assert sum(1 for s in shapes if s == (n_grid, 256, 64)) == 2  # no wide padding without the fold


@pytest.mark.gpu
@pytest.mark.parametrize("graph_name,channels,layers", [("o32_ico2", 128, 2), ("o96_ico5", 512, 2)])
def test_mapper_fold_on_the_kernels(graph_name, channels, layers, monkeypatch):
    """bf16 on the HIP kernels: folded and unfolded mappers both meet the f32 oracle (5e-2 of the output scale, the bf16
    bound of this suite), and agree with each other to bf16 rounding."""
    from anemoi_models_amd.graphs.synthetic import build_graph

    graph = build_graph(graph_name)
    monkeypatch.setenv("ANEMOI_AMD_DTYPE", "bf16")
    model, idx = _build(graph, channels, layers)
    x = torch.randn(1, 2, 1, graph["data"].num_nodes, idx.num_input, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        want = _oracle(model, graph, x, layers)
        model = model.to("cuda")
        monkeypatch.setenv("ANEMOI_AMD_EMBED_FOLD", "1")
        folded = model(x.cuda())
        monkeypatch.setenv("ANEMOI_AMD_EMBED_FOLD", "0")
        plain = model(x.cuda())
    e_fold, e_plain = rel_err(folded, want), rel_err(plain, want)
    print(f"{graph_name} {channels} ch: folded {e_fold:.2e}, unfolded {e_plain:.2e}, folded vs unfolded {rel_err(folded, plain):.2e}")
    assert e_fold < 5e-2 and e_plain < 5e-2
    assert e_fold < 2.0 * e_plain + 5e-3  # the fold must not cost accuracy (it removes two bf16 roundings)
    assert os.environ["ANEMOI_AMD_EMBED_FOLD"] == "0"
