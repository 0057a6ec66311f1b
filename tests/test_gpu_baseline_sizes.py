"""Parity at the BASELINE configurations' OWN sizes (north star: "outputs match the reference PyTorch-CPU forward on
identical random weights and a synthetic O96 atmospheric state within 1e-3 rel fp32").

* config 2: O96 -> ico-5, **all 16** GraphTransformer blocks, 512 channels, 16 heads -- f32 gated at 1e-3, bf16 reported
  against the f32 oracle with a stated bound;
* config 5: the same graph with 16 GNN blocks (pure edge-MLP message passing) -- f32 gated at 1e-3, bf16 with a bound;
* config 4 semantics at O96: a 2-step autoregressive rollout behind the interface (normaliser + model + advance_input);
* config 3 AT ITS OWN SIZE AND DEPTH (N320 -> ico-6, 542 080 grid rows, 1024 channels, 16 heads of 64, encoder in-degrees,
  K = 4096 reductions, all 16 processor blocks) against the oracle -- f32 gated at 1e-3 per variable, bf16 reported (round 6:
  the 2-block variant of this test is gone, the 16-block one covers the same code on the same shapes) -- and config 4 at
  N320 (4-step rollout, 16 blocks, bf16): interface rollout == chained forward +
  ``anemoi_advance_input``, state kept sharded over 2 ranks == unsharded.

The oracle (plain-PyTorch restatement, pinned to the reference by tests/test_oracle_golden.py) runs once per module on
the host cores (7-15 s per forward at these sizes).
"""

import numpy as np
import pytest
import torch

from oracle import reference_path as ref
from test_oracle_golden import graph_tensors, hier_graph_tensors

pytestmark = pytest.mark.gpu

DEV = "cuda"
N_PROG, N_FORC, N_DIAG = 20, 4, 2
BF16_BOUND = 1e-2  # bf16 storage / f32 accumulate over 16 residual blocks vs the f32 oracle (measured 2.2e-3 ... 3.0e-3, printed)


def rel_err(got, want):
    got, want = got.float().cpu(), want.float().cpu()
    return float((got - want).abs().max() / want.abs().max().clamp_min(1e-30))


def per_variable_rel_err(got, want):
    """SURVEY §8d parity gate: max over output variables of ||a - b||_inf / ||b||_inf."""
    got, want = got.float().cpu(), want.float().cpu()
    num = (got - want).abs().flatten(0, -2).max(dim=0).values
    den = want.abs().flatten(0, -2).max(dim=0).values.clamp_min(1e-30)
    return float((num / den).max())


def _make(processor):
    from anemoi_models_amd.graphs.synthetic import build_graph
    from anemoi_models_amd.models import AnemoiModelEncProcDec
    from anemoi_models_amd.utils.indices import SimpleDataIndices
    from anemoi_models_amd.utils.presets import model_config

    graph = build_graph("o96_ico5")
    idx = SimpleDataIndices(n_prognostic=N_PROG, n_forcing=N_FORC, n_diagnostic=N_DIAG)
    torch.manual_seed(1234)
    model = AnemoiModelEncProcDec(model_config=model_config(processor, 512, 16, 16), data_indices=idx, graph_data=graph)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("trainable"):
                p.normal_(0.0, 0.1)
    model.eval()
    x = torch.randn(1, 2, 1, graph["data"].num_nodes, idx.num_input, generator=torch.Generator().manual_seed(7))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        want = ref.model_forward(sd, graph_tensors(graph), x, num_heads=16, num_layers=16, num_chunks=2,
                                 prognostic_in=range(N_PROG), prognostic_out=range(N_PROG), processor=processor)
    return model.to(DEV), x.to(DEV), want, graph, idx


@pytest.fixture(scope="module")
def o96_gt():
    return _make("GraphTransformer")


@pytest.fixture(scope="module")
def o96_gnn():
    return _make("GNN")


def test_config2_o96_ico5_512ch_16_blocks_f32_vs_oracle(o96_gt, monkeypatch):
    monkeypatch.setenv("ANEMOI_AMD_DTYPE", "fp32")
    model, x, want, _, _ = o96_gt
    with torch.no_grad():
        got = model(x)
    err, err_v = rel_err(got, want), per_variable_rel_err(got, want)
    print(f"config 2 (O96 -> ico-5, 16 GT blocks, 512 ch) f32 vs CPU oracle: max rel {err:.3e}, per variable {err_v:.3e}")
    assert got.dtype == torch.float32 and got.shape == want.shape
    assert err < 1e-3 and err_v < 1e-3  # north-star gate


def test_config2_o96_ico5_512ch_16_blocks_bf16_vs_oracle(o96_gt, monkeypatch):
    monkeypatch.setenv("ANEMOI_AMD_DTYPE", "bf16")
    model, x, want, _, _ = o96_gt
    with torch.no_grad():
        got = model(x)
    err = rel_err(got, want)
    print(f"config 2 bf16 storage / f32 accumulate vs f32 CPU oracle: max rel {err:.3e} (bound {BF16_BOUND})")
    assert torch.isfinite(got).all() and err < BF16_BOUND


class _CudaAutocastPolicy(torch.overrides.TorchFunctionMode):
    """``torch.autocast("cpu", bfloat16)`` casts the Linear layers; CUDA autocast (what anemoi-training's ``bf16-mixed``
    applies to the reference) additionally runs ``layer_norm``, ``sum``, ``exp`` and ``softmax`` in f32 with f32 results
    (torch's CUDA autocast "fp32" op list).  This mode adds that part on the CPU, so that the oracle under both is the
    reference's production arithmetic: bf16 GEMMs and bf16 residual stream, f32 LayerNorm statistics, f32 attention
    scores / segment softmax / scatter sums.  ``AutocastLayerNorm`` (reference layers/utils.py:27-39, the MLP class's
    LayerNorm) casts back to its input type inside ``oracle.reference_path.mlp``."""

    _F32 = {torch.nn.functional.layer_norm, torch.layer_norm, torch.sum, torch.Tensor.sum, torch.exp, torch.Tensor.exp,
            torch.softmax, torch.Tensor.softmax, torch.nn.functional.softmax}

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in self._F32:
            up = lambda t: t.float() if isinstance(t, torch.Tensor) and t.dtype == torch.bfloat16 else t  # noqa: E731
            args = tuple(up(a) for a in args)
            kwargs = {k: up(v) for k, v in kwargs.items()}
        return func(*args, **kwargs)


def oracle_under_bf16_autocast(fn):
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16), _CudaAutocastPolicy():
        return fn()


def test_config2_bf16_anchored_to_the_oracle_under_bf16_autocast(o96_gt, monkeypatch):
    """What does bf16 cost the REFERENCE?  The oracle under bf16 autocast (f32 weights, bf16 GEMMs / activations, f32
    LayerNorm and softmax as CUDA autocast does) against the f32 oracle is the yardstick; the HIP bf16 path (bf16 storage,
    f32 accumulation) must not be worse than 1.5 x that, on the prediction AND on the encoder output (mesh latent)."""
    import bench

    model, x, want, graph, _ = o96_gt
    sd = {k: (v.detach().float() if v.is_floating_point() else v.detach()).cpu() for k, v in model.state_dict().items()}
    kw = dict(num_heads=16, num_layers=16, num_chunks=2, prognostic_in=range(N_PROG), prognostic_out=range(N_PROG),
              return_stages=True)
    gt = graph_tensors(graph)
    with torch.no_grad():
        want32, st32 = ref.model_forward(sd, gt, x.cpu(), **kw)
    assert torch.equal(want32, want)
    auto, st_auto = oracle_under_bf16_autocast(lambda: ref.model_forward(sd, gt, x.cpu(), **kw))
    assert st_auto["x_latent"].dtype == torch.bfloat16  # the autocast really reached the residual stream
    ref_out, ref_lat = rel_err(auto, want32), rel_err(st_auto["x_latent"], st32["x_latent"])
    monkeypatch.setenv("ANEMOI_AMD_DTYPE", "bf16")
    got, latent = bench.device_forward_with_latent(model, x)
    hip_out, hip_lat = rel_err(got, want32), rel_err(latent, st32["x_latent"])
    print(f"config 2, bf16 vs the f32 oracle -- prediction: HIP {hip_out:.3e}, oracle under bf16 autocast {ref_out:.3e}; "
          f"encoder latent: HIP {hip_lat:.3e}, oracle under bf16 autocast {ref_lat:.3e}")
    if not (hip_out <= 1.5 * ref_out and hip_lat <= 1.5 * ref_lat):
        # one whole-suite run of round 6 measured 1.21e-2 here where every other run of the same library measures 6.50e-3
        # (DESIGN section 2): say whether the SAME model gives the same bits when asked again -- a persistent state or a
        # single forward that went wrong
        again = [bench.device_forward_with_latent(model, x) for _ in range(3)]
        lat32 = st32["x_latent"]
        pytest.fail(f"prediction {hip_out:.4e} (oracle under autocast {ref_out:.4e}), latent {hip_lat:.4e} ({ref_lat:.4e}); "
                    f"three more forwards of the same model: latent "
                    f"{[round(rel_err(l, lat32), 6) for _, l in again]}, bit-equal to the failing one "
                    f"{[bool(torch.equal(l, latent) and torch.equal(y, got)) for y, l in again]}; latent elements that "
                    f"moved against the first of them: {int((again[0][1] != latent).sum())} in "
                    f"{int((again[0][1] != latent).any(1).sum())} rows")


def test_config5_gnn_o96_512ch_16_blocks_f32_vs_oracle(o96_gnn, monkeypatch):
    monkeypatch.setenv("ANEMOI_AMD_DTYPE", "fp32")
    model, x, want, _, _ = o96_gnn
    with torch.no_grad():
        got = model(x)
    err, err_v = rel_err(got, want), per_variable_rel_err(got, want)
    print(f"config 5 (O96, 16 GNN blocks, 512 ch) f32 vs CPU oracle: max rel {err:.3e}, per variable {err_v:.3e}")
    assert err < 1e-3 and err_v < 1e-3


def test_config5_gnn_o96_512ch_16_blocks_bf16_vs_oracle(o96_gnn, monkeypatch):
    """The 512-channel bf16 route of the GNN processor (persistent GEMM, 16-byte-lane gather_add_act, segment_sum)."""
    monkeypatch.setenv("ANEMOI_AMD_DTYPE", "bf16")
    model, x, want, _, _ = o96_gnn
    with torch.no_grad():
        got = model(x)
    err = rel_err(got, want)
    print(f"config 5 bf16 vs f32 CPU oracle: max rel {err:.3e} (bound {BF16_BOUND})")
    assert torch.isfinite(got).all() and err < BF16_BOUND


@pytest.fixture(scope="module")
def o96_tfm():
    return _make("Transformer")


def test_config2_size_transformer_processor_16_blocks_f32_and_bf16_vs_oracle(o96_tfm, monkeypatch):
    """The Transformer-processor family at BASELINE config 2's size (O96 -> ico-5, 512 channels, 16 blocks of mesh-node
    MultiHeadSelfAttention over all 10 242 nodes, 16 heads of 32; reference layers/attention.py:67-112, layers/block.py:99-105)
    against ``oracle.model_forward(processor="Transformer")``: f32 (exact-f32 GEMMs, the f32 attention route) gated at 1e-3 per
    output variable; bf16 (the MFMA flash attention, D = 32) bounded.  Round 6: until now this family was held to the oracle at
    config 1's size only (golden vectors) and to an f64 formula per attention call at mesh size."""
    model, x, want, _, _ = o96_tfm
    monkeypatch.setenv("ANEMOI_AMD_DTYPE", "fp32")
    with torch.no_grad():
        got = model(x)
    err, err_v = rel_err(got, want), per_variable_rel_err(got, want)
    print(f"config-2 size, 16 Transformer blocks, f32 vs CPU oracle: max rel {err:.3e}, per variable {err_v:.3e}")
    assert got.dtype == torch.float32 and got.shape == want.shape
    assert err < 1e-3 and err_v < 1e-3
    monkeypatch.setenv("ANEMOI_AMD_DTYPE", "bf16")
    with torch.no_grad():
        got16 = model(x)
    e16 = rel_err(got16, want)
    print(f"config-2 size, 16 Transformer blocks, bf16 vs f32 CPU oracle: max rel {e16:.3e} (bound {BF16_BOUND})")
    assert torch.isfinite(got16).all() and e16 < BF16_BOUND


def test_rollout_2_steps_o96_vs_oracle(monkeypatch):
    """BASELINE config 4 semantics at O96 / 512 ch / 16 blocks: 2 autoregressive steps through
    ``AnemoiModelInterface.rollout`` (normaliser folded into the first / last kernel, ``anemoi_advance_input`` between the
    steps) against ``oracle.rollout`` (each step the pinned ``model_forward`` + the reference normaliser arithmetic)."""
    from anemoi_models_amd.graphs.synthetic import build_graph
    from anemoi_models_amd.interface import AnemoiModelInterface
    from anemoi_models_amd.utils.indices import SimpleDataIndices
    from anemoi_models_amd.utils.presets import model_config

    monkeypatch.setenv("ANEMOI_AMD_DTYPE", "fp32")
    graph = build_graph("o96_ico5")
    n_prog, n_forc, n_diag = 10, 2, 1
    n_all = n_prog + n_forc + n_diag
    cfg = model_config("GraphTransformer", 512, 16, 16)
    cfg["data"] = {"forcing": [f"forc_{i}" for i in range(n_forc)], "diagnostic": ["diag_0"],
                   "processors": {"normalizer": {"_target_": "anemoi.models.preprocessing.normalizer.InputNormalizer",
                                                 "config": {"default": "mean-std", "min-max": ["prog_3"],
                                                            "max": ["prog_4"], "none": ["forc_0"]}}}}
    cfg["model"]["model"] = {"_target_": "anemoi.models.models.encoder_processor_decoder.AnemoiModelEncProcDec"}
    gen = torch.Generator().manual_seed(11)
    mean = (torch.randn(n_all, generator=gen) * 3.0).numpy().astype(np.float32)
    stdev = (0.5 + torch.rand(n_all, generator=gen) * 2.0).numpy().astype(np.float32)
    stats = {"mean": mean, "stdev": stdev, "minimum": mean - 3.0 * stdev, "maximum": mean + 3.5 * stdev}
    idx = SimpleDataIndices(n_prognostic=n_prog, n_forcing=n_forc, n_diagnostic=n_diag)
    torch.manual_seed(1234)
    iface = AnemoiModelInterface(config=type(cfg)(cfg), graph_data=graph, statistics=stats, data_indices=idx, metadata={})
    with torch.no_grad():
        for name, p in iface.named_parameters():
            if name.endswith("trainable"):
                p.normal_(0.0, 0.1)
    iface.eval()
    n_grid = graph["data"].num_nodes
    in_idx = idx.data.input.full.long()
    z = torch.randn((1, 2, n_grid, n_prog + n_forc), generator=torch.Generator().manual_seed(7))
    batch = z * torch.from_numpy(stdev)[in_idx] + torch.from_numpy(mean)[in_idx]
    f_in = idx.internal_model.input.forcing.long()
    f_data = in_idx[f_in]
    zf = torch.randn((2, 1, n_grid, n_forc), generator=torch.Generator().manual_seed(8))
    forcings = zf * torch.from_numpy(stdev)[f_data] + torch.from_numpy(mean)[f_data]
    sd = {k: v.clone() for k, v in iface.state_dict().items()}
    with torch.no_grad():
        want = ref.rollout(sd, graph_tensors(graph), batch, 2, forcings, multi_step=2, prognostic_in=range(n_prog),
                           prognostic_out=range(n_prog), forcing_in=f_in.tolist(), num_heads=16, num_layers=16,
                           num_chunks=2)
    iface = iface.to(DEV)
    got = iface.rollout(batch.to(DEV), 2, forcings.to(DEV))
    assert got.shape == want.shape
    e1, e2 = per_variable_rel_err(got[0], want[0]), per_variable_rel_err(got[1], want[1])
    print(f"O96 rollout, f32, per-variable rel err: step 1 {e1:.3e}, step 2 {e2:.3e}")
    assert e1 < 1e-3 and e2 < 1e-3


@pytest.mark.parametrize("processor", ["GraphTransformer", "GNN", "Transformer"])
def test_forward_is_bit_reproducible_under_repetition(processor, monkeypatch):
    """No kernel of the path uses atomics, so two runs of one input must agree bit for bit; a difference is a race (the
    round-2 attention bug showed exactly this way).  BASELINE config 2 sizes, bf16, shifting allocations between the runs."""
    import random
    import sys

    sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
    import bench

    monkeypatch.setenv("ANEMOI_AMD_DTYPE", "bf16")
    model, _graph, x, _ = bench.build("cfg2", torch.device("cuda", 0), processor)
    model.eval()
    random.seed(2)
    first = None
    for it in range(12):
        junk = [torch.full((random.randint(1, 1 << 21),), float("nan"), device="cuda") for _ in range(random.randint(0, 3))]
        with torch.no_grad():
            y = model(x)
        del junk
        assert bool(torch.isfinite(y).all())
        if first is None:
            first = y.clone()
        else:
            assert torch.equal(first, y), f"run {it} differs from run 0"


# ------------------------------------------------------------------------------------------- config 3 / 4 at N320 size
def test_config3_n320_ico6_1024ch_all_16_blocks_f32_vs_oracle(monkeypatch):
    """The north star's parity statement at the metric's OWN size and depth: N320 -> ico-6, 1024 channels, ALL 16
    GraphTransformer blocks, f32 (the exact-f32 MFMA route), against ``oracle.model_forward`` on the same weights / input:
    <= 1e-3 on the prediction (max and per output variable) and on the encoder latent; bf16 (the headline dtype) reported
    next to it, bounded.  ~150 s of oracle on the GPU box's cores, the longest test of the suite (the 1e-3 budget
    is "tightest in config #3 with 16 residual blocks"); ``bench.py``'s default line repeats it (``cpu_baseline.parity_fp32``)."""
    import bench

    model, graph, x, idx = bench.build("cfg3", "cpu")
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        want, stages = ref.model_forward(sd, graph_tensors(graph), x, num_heads=16, num_layers=16, num_chunks=2,
                                         prognostic_in=range(80), prognostic_out=range(80), mapper_chunks=8,
                                         return_stages=True)
    del sd
    model, x = model.to(DEV), x.to(DEV)
    monkeypatch.setenv("ANEMOI_AMD_DTYPE", "fp32")
    got, latent = bench.device_forward_with_latent(model, x)
    err, err_v, err_l = rel_err(got, want), per_variable_rel_err(got, want), rel_err(latent, stages["x_latent"])
    print(f"config 3 (N320 -> ico-6, 1024 ch, 16 GT blocks) f32 vs CPU oracle: max rel {err:.3e}, per variable {err_v:.3e}, "
          f"encoder latent {err_l:.3e}")
    assert got.dtype == torch.float32 and got.shape == want.shape == (1, 1, 542080, 80)
    assert err < 1e-3 and err_v < 1e-3 and err_l < 1e-3  # north-star gate: own size, own depth
    monkeypatch.setenv("ANEMOI_AMD_DTYPE", "bf16")
    got16, latent16 = bench.device_forward_with_latent(model, x)
    e16, e16_l = rel_err(got16, want), rel_err(latent16, stages["x_latent"])
    print(f"config 3, 16 blocks, bf16 storage / f32 accumulate vs f32 CPU oracle: max rel {e16:.3e}, encoder latent {e16_l:.3e}")
    assert torch.isfinite(got16).all() and e16 < BF16_BOUND and e16_l < bench.LATENT_BOUND["bf16"]


def test_expand_edges_on_device_bit_exact(golden_index_ops):
    """SURVEY section 8 a3 on DEVICE tensors: ``runtime.expand_edges`` (reference layers/mapper.py:150-171) against the
    vectors recorded from the reference (int64, bit exact)."""
    from anemoi_models_amd import runtime

    z = golden_index_ops
    inc = torch.tensor([[70], [31]], dtype=torch.int64, device=DEV)
    got = runtime.expand_edges(z["expand.edge_index"].to(DEV), inc, 3)
    assert got.is_cuda and got.dtype == torch.int64 and torch.equal(got.cpu(), z["expand.out"])


def test_config4_n320_rollout_4_steps_16_blocks_bf16(monkeypatch):
    """BASELINE config 4 at its own size (N320, 16 blocks, 1024 ch, bf16, 4 lead times) behind
    ``AnemoiModelInterface.rollout`` (reference interface/__init__.py:97-123 + anemoi-training's ``advance_input``):
    (i) the rollout loop == four chained ``pre-process -> model.forward -> post-process`` calls with
    ``anemoi_advance_input`` in between, BIT FOR BIT (generic route on both sides); (ii) the route with the normaliser
    fused into the first / last kernel agrees with it to bf16 rounding; (iii) every lead time is finite and the state
    really advances (step k differs from step k - 1)."""
    from anemoi_models_amd import ops
    from anemoi_models_amd.graphs.synthetic import build_graph
    from anemoi_models_amd.interface import AnemoiModelInterface
    from anemoi_models_amd.utils.indices import SimpleDataIndices
    from anemoi_models_amd.utils.presets import model_config

    monkeypatch.setenv("ANEMOI_AMD_DTYPE", "bf16")
    n_steps = 4
    graph = build_graph("n320_ico6")
    n_prog, n_forc = 80, 10
    n_all = n_prog + n_forc
    cfg = model_config("GraphTransformer", 1024, 16, 16)
    cfg["data"] = {"forcing": [f"forc_{i}" for i in range(n_forc)], "diagnostic": [],
                   "processors": {"normalizer": {"_target_": "anemoi.models.preprocessing.normalizer.InputNormalizer",
                                                 "config": {"default": "mean-std", "min-max": ["prog_3"],
                                                            "max": ["prog_4"], "none": ["forc_0"]}}}}
    cfg["model"]["model"] = {"_target_": "anemoi.models.models.encoder_processor_decoder.AnemoiModelEncProcDec"}
    gen = torch.Generator().manual_seed(11)
    mean = (torch.randn(n_all, generator=gen) * 3.0).numpy().astype(np.float32)
    stdev = (0.5 + torch.rand(n_all, generator=gen) * 2.0).numpy().astype(np.float32)
    stats = {"mean": mean, "stdev": stdev, "minimum": mean - 3.0 * stdev, "maximum": mean + 3.5 * stdev}
    idx = SimpleDataIndices(n_prognostic=n_prog, n_forcing=n_forc, n_diagnostic=0)
    torch.manual_seed(1234)
    with torch.device(DEV):
        iface = AnemoiModelInterface(config=type(cfg)(cfg), graph_data=graph.to(DEV), statistics=stats, data_indices=idx,
                                     metadata={})
    with torch.no_grad():
        for name, p in iface.named_parameters():
            if name.endswith("trainable"):
                p.normal_(0.0, 0.1)
    iface = iface.to(DEV).eval()
    n_grid = graph["data"].num_nodes
    in_idx = idx.data.input.full.long()
    z = torch.randn((1, 2, n_grid, n_all), generator=torch.Generator().manual_seed(7))
    batch = (z * torch.from_numpy(stdev)[in_idx] + torch.from_numpy(mean)[in_idx]).to(DEV)
    f_in = idx.internal_model.input.forcing.long()
    f_data = in_idx[f_in]
    zf = torch.randn((n_steps, 1, n_grid, n_forc), generator=torch.Generator().manual_seed(8))
    forcings = (zf * torch.from_numpy(stdev)[f_data] + torch.from_numpy(mean)[f_data]).to(DEV)

    monkeypatch.setenv("ANEMOI_AMD_FUSE_NORMALIZER", "0")
    got = iface.rollout(batch, n_steps, forcings)
    # the chain, written out: reference predict_step pieces + advance_input on the normalised state
    cmap = iface._advance_map(batch.device)
    f_idx = f_in.to(DEV)
    with torch.no_grad():
        x = iface.pre_processors(batch, in_place=False)[:, 0:iface.multi_step, None, ...].float().contiguous().clone()
        chain = []
        for step in range(n_steps):
            y_hat = iface.model(x)
            chain.append(iface.post_processors(y_hat, in_place=False))
            if step + 1 == n_steps:
                break
            full = x[:, -1, 0].clone()
            full[..., f_idx] = forcings[step].to(full)
            f_norm = iface.pre_processors(full[:, None], in_place=False)[:, 0][..., f_idx][:, None].contiguous()
            ops.advance_input(x, y_hat.float().contiguous(), cmap, f_norm)
        chain = torch.stack(chain)
    assert got.shape == chain.shape == (n_steps, 1, 1, n_grid, n_prog)
    assert torch.isfinite(got).all()
    assert torch.equal(got, chain)  # (i)
    for k in range(1, n_steps):  # (iii)
        assert float((got[k] - got[k - 1]).abs().max()) > 0
    monkeypatch.delenv("ANEMOI_AMD_FUSE_NORMALIZER")
    fused = iface.rollout(batch, n_steps, forcings)  # (ii)
    errs = [per_variable_rel_err(fused[k], got[k]) for k in range(n_steps)]
    print("config 4 (N320, 16 blocks, bf16, 4 lead times): fused-normaliser vs generic route, per-variable rel err per "
          "lead time: " + ", ".join(f"{e:.2e}" for e in errs))
    assert max(errs) < BF16_BOUND


def test_config4_n320_rollout_sharded_state_ranks_sharing_one_gpu(tmp_path):
    """BASELINE config 4 on the N > 1 route at N320 size (16 blocks, 1024 ch, bf16, 4 lead times), stepped exactly as
    ``bench.py --rollout 4 --gpus N`` steps it: the state stays SHARDED between the lead times (grid-halo all-to-all-v, no
    per-step all-gather), against the single-device chain.  Two ranks as processes sharing cuda:0,
    collectives host-staged over gloo (the test box has one GPU)."""
    import os
    import subprocess
    import sys

    world, steps = 2, 4
    port = 29400 + (os.getpid() % 200)
    out = str(tmp_path / "res")
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_gpu_shared_rollout.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), str(port), out, "cfg3", str(steps)])
             for r in range(world)]
    try:
        codes = [p.wait(timeout=1500) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
    assert codes == [0] * world
    infos = [torch.load(f"{out}.{r}") for r in range(world)]
    assert sum(i["own"] for i in infos) == 40962
    for i in infos:
        print(f"config 4 sharded state, world {world}: max |sharded - unsharded| {i['err']:.3e} at output scale {i['scale']:.3e}, "
              f"grid halo {i['grid_halo']} of {i['grid']} rows")
        assert i["finite"] and i["shape"] == (1, 1, 542080, 80)
        assert 0 < i["grid_halo"] < i["grid"] // 4
        assert i["err"] <= BF16_BOUND * max(1.0, i["scale"]), i


def test_hierarchical_model_o96_three_levels_f32_and_bf16_vs_oracle_and_a_bf16_training_step(monkeypatch):
    """The hierarchical model (reference models/hierarchical.py:178-308) beyond its golden O32 vectors:
    O96 grid -> ico-5 (256 ch) -> ico-4 (512 ch) -> ico-3 (1024 ch), level processors of 2 blocks down and up, heads of
    16 / 32 / 64 -- the channel widths and head sizes of configs 2 and 3 in one model.  f32 gated at 1e-3 per variable against
    ``oracle.hierarchical_forward``, bf16 within the stated bound; one bf16 training step (differentiable route at these
    widths) against the f32 step of the same weights."""
    from anemoi_models_amd.graphs.synthetic import build_hierarchical_graph
    from anemoi_models_amd.models import AnemoiModelEncProcDecHierarchical
    from anemoi_models_amd.utils.indices import SimpleDataIndices
    from anemoi_models_amd.utils.presets import hierarchical_model_config

    hidden = ["hidden_1", "hidden_2", "hidden_3"]
    graph = build_hierarchical_graph("o96", (5, 4, 3))
    idx = SimpleDataIndices(n_prognostic=N_PROG, n_forcing=N_FORC, n_diagnostic=N_DIAG)
    torch.manual_seed(4321)
    model = AnemoiModelEncProcDecHierarchical(model_config=hierarchical_model_config(256, 16, hidden=hidden), data_indices=idx,
                                              graph_data=graph)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("trainable"):
                p.normal_(0.0, 0.1)
    model.eval()
    x = torch.randn(1, 2, 1, graph["data"].num_nodes, idx.num_input, generator=torch.Generator().manual_seed(9))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        want = ref.hierarchical_forward(sd, hier_graph_tensors(graph, hidden), x, hidden=hidden, num_heads=16, level_layers=2,
                                        prognostic_in=list(range(N_PROG)), prognostic_out=list(range(N_PROG)))
    model, x = model.to(DEV), x.to(DEV)
    res = {}
    for mode in ("fp32", "bf16"):
        monkeypatch.setenv("ANEMOI_AMD_DTYPE", mode)
        with torch.no_grad():
            res[mode] = model(x)
    err, err_v, err_b = rel_err(res["fp32"], want), per_variable_rel_err(res["fp32"], want), rel_err(res["bf16"], want)
    print(f"hierarchical O96 -> ico-5 / 4 / 3 (256 / 512 / 1024 ch) f32 vs CPU oracle: max rel {err:.3e}, per variable "
          f"{err_v:.3e}; bf16: {err_b:.3e} (bound {BF16_BOUND})")
    assert res["fp32"].shape == want.shape and err < 1e-3 and err_v < 1e-3
    assert torch.isfinite(res["bf16"]).all() and err_b < BF16_BOUND
    # one training step per precision on the same weights
    model.train()
    dy = torch.randn(want.shape, generator=torch.Generator().manual_seed(3)).to(DEV)
    grads = {}
    for mode in ("fp32", "bf16"):
        monkeypatch.setenv("ANEMOI_AMD_DTYPE", mode)
        model.zero_grad(set_to_none=True)
        y = model(x)
        assert y.requires_grad and rel_err(y.detach(), want) < (1e-3 if mode == "fp32" else BF16_BOUND)
        y.backward(dy)
        grads[mode] = {k: p.grad.float().clone() for k, p in model.named_parameters() if p.grad is not None}
    assert set(grads["bf16"]) == set(grads["fp32"]) and any(k.startswith("upscale.") for k in grads["fp32"])
    scale_all = max(float(g.abs().max()) for g in grads["fp32"].values())
    worst = 0.0
    for k, g32 in grads["fp32"].items():
        e_k = float((grads["bf16"][k] - g32).abs().max()) / max(float(g32.abs().max()), 0.05 * scale_all)
        worst = max(worst, e_k)
        assert e_k <= 8e-2, (k, e_k)
    print(f"bf16 training step against the f32 step: worst parameter gradient {worst:.3e} of its scale")


def test_all_gnn_model_o96_512ch_f32_and_bf16_vs_oracle_and_a_bf16_training_step(monkeypatch):
    """GNN mappers + GNN processor (reference layers/mapper.py:421-705) beyond the golden O32
    vectors: config 5's graph and width (O96 -> ico-5, 512 channels), 4 processor blocks to bound the oracle's time.  f32
    gated at 1e-3 per variable, bf16 within the stated bound, one bf16 training step against the f32 step."""
    from anemoi_models_amd.graphs.synthetic import build_graph
    from anemoi_models_amd.models import AnemoiModelEncProcDec
    from anemoi_models_amd.utils.indices import SimpleDataIndices
    from anemoi_models_amd.utils.presets import model_config

    graph = build_graph("o96_ico5")
    idx = SimpleDataIndices(n_prognostic=N_PROG, n_forcing=N_FORC, n_diagnostic=N_DIAG)
    torch.manual_seed(1234)
    model = AnemoiModelEncProcDec(model_config=model_config("GNN", 512, 4, 16, mappers="GNN"), data_indices=idx,
                                  graph_data=graph)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("trainable"):
                p.normal_(0.0, 0.1)
    model.eval()
    x = torch.randn(1, 2, 1, graph["data"].num_nodes, idx.num_input, generator=torch.Generator().manual_seed(7))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        want = ref.model_forward(sd, graph_tensors(graph), x, num_heads=16, num_layers=4, num_chunks=2,
                                 prognostic_in=range(N_PROG), prognostic_out=range(N_PROG), processor="GNN", mappers="GNN")
    model, x = model.to(DEV), x.to(DEV)
    res = {}
    for mode in ("fp32", "bf16"):
        monkeypatch.setenv("ANEMOI_AMD_DTYPE", mode)
        with torch.no_grad():
            res[mode] = model(x)
    err, err_v, err_b = rel_err(res["fp32"], want), per_variable_rel_err(res["fp32"], want), rel_err(res["bf16"], want)
    print(f"all-GNN model at config 5's size (4 blocks) f32 vs CPU oracle: max rel {err:.3e}, per variable {err_v:.3e}; "
          f"bf16: {err_b:.3e} (bound {BF16_BOUND})")
    assert res["fp32"].shape == want.shape and err < 1e-3 and err_v < 1e-3
    assert torch.isfinite(res["bf16"]).all() and err_b < BF16_BOUND
    model.train()
    dy = torch.randn(want.shape, generator=torch.Generator().manual_seed(3)).to(DEV)
    grads = {}
    for mode in ("fp32", "bf16"):
        monkeypatch.setenv("ANEMOI_AMD_DTYPE", mode)
        model.zero_grad(set_to_none=True)
        y = model(x)
        assert y.requires_grad and rel_err(y.detach(), want) < (1e-3 if mode == "fp32" else BF16_BOUND)
        y.backward(dy)
        grads[mode] = {k: p.grad.float().clone() for k, p in model.named_parameters() if p.grad is not None}
    assert set(grads["bf16"]) == set(grads["fp32"]) and any(k.startswith("encoder.") for k in grads["fp32"])
    scale_all = max(float(g.abs().max()) for g in grads["fp32"].values())
    worst = max(float((grads["bf16"][k] - g32).abs().max()) / max(float(g32.abs().max()), 0.05 * scale_all)
                for k, g32 in grads["fp32"].items())
    print(f"bf16 training step against the f32 step: worst parameter gradient {worst:.3e} of its scale")
    assert worst <= 8e-2

