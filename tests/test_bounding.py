"""Output boundings (reference layers/bounding.py:60-124).

* the reference's own exact-value cases (reference tests/layers/test_bounding.py:35-105) through the module route on CPU
  and, on the GPU, through BOTH routes (module + the compiled op list on ``anemoi_bound_output``);
* a config-1 model and interface with a ``bounding:`` list against vectors recorded from the real reference
  (tests/golden/make_golden.py::golden_bounding): oracle, host wiring (CPU), HIP path (GPU).
"""

import math

import pytest
import torch

import _cpu_ops
from conftest import load_npz
from conftest import split_prefix
from anemoi_models_amd.layers.bounding import FractionBounding
from anemoi_models_amd.layers.bounding import HardtanhBounding
from anemoi_models_amd.layers.bounding import ReluBounding
from anemoi_models_amd.layers.bounding import bounded_columns
from anemoi_models_amd.layers.bounding import compile_boundings
from anemoi_models_amd.models import AnemoiModelEncProcDec
from anemoi_models_amd.utils.config import instantiate
from anemoi_models_amd.utils.indices import SimpleDataIndices
from anemoi_models_amd.utils.presets import model_config
from oracle import reference_path as ref
from test_oracle_golden import graph_tensors

VARIABLES = ["var1", "var2"]
NAME_TO_INDEX = {"var1": 0, "var2": 1, "total_var": 2}
BOUNDING = [  # as in tests/golden/make_golden.py
    {"_target_": "anemoi.models.layers.bounding.ReluBounding", "variables": ["prog_0", "diag_0"]},
    {"_target_": "anemoi.models.layers.bounding.HardtanhBounding", "variables": ["prog_1", "prog_0"], "min_val": -0.5,
     "max_val": 0.75},
    {"_target_": "anemoi.models.layers.bounding.FractionBounding", "variables": ["prog_3", "prog_2"], "min_val": 0.0,
     "max_val": 1.0, "total_var": "prog_4"},
]


def input_tensor():
    return torch.tensor([[-1.0, 2.0, 3.0], [4.0, -5.0, 6.0], [0.5, 0.5, 0.5]])


def reference_cases():
    """(boundings, expected output) of reference tests/layers/test_bounding.py:35-76."""
    relu = ReluBounding(variables=VARIABLES, name_to_index=NAME_TO_INDEX)
    tanh = HardtanhBounding(variables=VARIABLES, name_to_index=NAME_TO_INDEX, min_val=-1.0, max_val=1.0)
    frac = FractionBounding(variables=VARIABLES, name_to_index=NAME_TO_INDEX, min_val=0.0, max_val=1.0,
                            total_var="total_var")
    relu1 = ReluBounding(variables=VARIABLES[:-1], name_to_index=NAME_TO_INDEX)
    tanh2 = HardtanhBounding(variables=VARIABLES, name_to_index=NAME_TO_INDEX, min_val=0.5, max_val=1.75)
    return [
        ([relu], torch.tensor([[0.0, 2.0, 3.0], [4.0, 0.0, 6.0], [0.5, 0.5, 0.5]])),
        ([tanh], torch.tensor([[-1.0, 1.0, 3.0], [1.0, -1.0, 6.0], [0.5, 0.5, 0.5]])),
        ([frac], torch.tensor([[0.0, 3.0, 3.0], [6.0, 0.0, 6.0], [0.25, 0.25, 0.5]])),
        ([relu1], torch.tensor([[0.0, 2.0, 3.0], [4.0, -5.0, 6.0], [0.5, 0.5, 0.5]])),
        ([relu1, tanh2], torch.tensor([[0.5, 1.75, 3.0], [1.75, 0.5, 6.0], [0.5, 0.5, 0.5]])),
    ]


def op_tensors(op_list, device):
    return (torch.tensor([o[0] for o in op_list], dtype=torch.int32, device=device),
            torch.tensor([o[1] for o in op_list], dtype=torch.float32, device=device),
            torch.tensor([o[2] for o in op_list], dtype=torch.float32, device=device),
            torch.tensor([o[3] for o in op_list], dtype=torch.int32, device=device))


# ------------------------------------------------------------------------------------------------------ CPU
@pytest.mark.parametrize("case", range(5))
def test_reference_exact_value_cases_module_route(case):
    boundings, want = reference_cases()[case]
    x = input_tensor()
    for b in boundings:
        x = b(x)
    assert torch.equal(x, want)


@pytest.mark.parametrize("case", range(5))
def test_reference_exact_value_cases_compiled_op_list(case):
    """The op list the model hands to anemoi_bound_output, executed by its sequential CPU restatement."""
    boundings, want = reference_cases()[case]
    op_list = compile_boundings(boundings)
    assert op_list is not None
    got = _cpu_ops.bound_output(input_tensor(), *op_tensors(op_list, "cpu"))
    assert torch.equal(got, want)


def test_hydra_style_instantiation_of_reference_targets():
    """reference tests/layers/test_bounding.py:79-105: the reference's ``_target_`` strings resolve to this package."""
    defs = [
        {"_target_": "anemoi.models.layers.bounding.ReluBounding", "variables": VARIABLES},
        {"_target_": "anemoi.models.layers.bounding.HardtanhBounding", "variables": VARIABLES, "min_val": 0.0,
         "max_val": 1.0},
        {"_target_": "anemoi.models.layers.bounding.FractionBounding", "variables": VARIABLES, "min_val": 0.0,
         "max_val": 1.0, "total_var": "total_var"},
    ]
    kinds = [ReluBounding, HardtanhBounding, FractionBounding]
    for d, kind in zip(defs, kinds):
        b = instantiate(d, name_to_index=NAME_TO_INDEX)
        assert type(b) is kind
        b(input_tensor())
    with pytest.raises(AssertionError):  # unknown variable: the reference's index builder asserts
        ReluBounding(variables=["nope"], name_to_index=NAME_TO_INDEX)


def test_compiled_ops_equal_module_chain_on_random_data():
    """Duplicated variables, a total that is itself bounded, NaN / inf inputs: op list == chained modules, bit for bit."""
    n2i = {f"v{i}": i for i in range(7)}
    chain = [
        ReluBounding(variables=["v5", "v0", "v5"], name_to_index=n2i),
        FractionBounding(variables=["v1", "v2", "v3"], name_to_index=n2i, min_val=0.0, max_val=1.0, total_var="v2"),
        HardtanhBounding(variables=["v0", "v3"], name_to_index=n2i, min_val=-0.25, max_val=0.5),
        FractionBounding(variables=["v4"], name_to_index=n2i, min_val=-1.0, max_val=1.0, total_var="v0"),
    ]
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 1, 500, 7, generator=g) * 2
    x[0, 0, 0, 1] = float("nan")
    x[0, 0, 1, 5] = float("inf")
    x[0, 0, 2, 0] = -float("inf")
    want = x.clone()
    for b in chain:
        want = b(want)
    op_list = compile_boundings(chain)
    assert bounded_columns(op_list) == [0, 1, 2, 3, 4, 5]
    got = _cpu_ops.bound_output(x.clone(), *op_tensors(op_list, "cpu"))
    assert torch.equal(torch.nan_to_num(got, nan=123.0), torch.nan_to_num(want, nan=123.0))
    assert math.isnan(float(got[0, 0, 0, 1]))

    class Custom(ReluBounding):
        pass

    assert compile_boundings([Custom(variables=["v0"], name_to_index=n2i)]) is None  # unknown class: module route


def test_oracle_model_with_boundings_vs_reference(graph_o32, golden_cfg1_gt):
    gold = golden_cfg1_gt
    idx = SimpleDataIndices(n_prognostic=10, n_forcing=2, n_diagnostic=1)
    y = ref.model_forward(split_prefix(gold, "sd."), graph_tensors(graph_o32), gold["x"], num_heads=16, num_layers=4,
                          num_chunks=2, prognostic_in=range(10), prognostic_out=range(10), boundings=BOUNDING,
                          name_to_index_out=idx.internal_model.output.name_to_index)
    torch.testing.assert_close(y, load_npz("bounding_gt.npz")["y"], atol=1e-4, rtol=1e-4)


def bounded_model(graph):
    cfg = model_config("GraphTransformer", 64, 4, 16)
    cfg["model"]["bounding"] = [dict(b) for b in BOUNDING]
    idx = SimpleDataIndices(n_prognostic=10, n_forcing=2, n_diagnostic=1)
    return AnemoiModelEncProcDec(model_config=type(cfg)(cfg), data_indices=idx, graph_data=graph)


def bounded_interface(graph, gold):
    from anemoi_models_amd.interface import AnemoiModelInterface
    from test_host_logic import NORMALIZER_METHODS

    cfg = model_config("GraphTransformer", 64, 4, 16)
    cfg["model"]["bounding"] = [dict(b) for b in BOUNDING]
    cfg["data"] = {"forcing": ["forc_0", "forc_1"], "diagnostic": ["diag_0"],
                   "processors": {"normalizer": {"_target_": "anemoi.models.preprocessing.normalizer.InputNormalizer",
                                                 "config": dict(NORMALIZER_METHODS)}}}
    cfg["model"]["model"] = {"_target_": "anemoi.models.models.encoder_processor_decoder.AnemoiModelEncProcDec"}
    stats = {k: v.numpy() for k, v in split_prefix(gold, "stat.").items()}
    idx = SimpleDataIndices(n_prognostic=10, n_forcing=2, n_diagnostic=1)
    return AnemoiModelInterface(config=type(cfg)(cfg), graph_data=graph, statistics=stats, data_indices=idx, metadata={})


def test_model_with_bounding_list_host_wiring(graph_o32, golden_cfg1_gt, monkeypatch):
    _cpu_ops.install(monkeypatch)
    model = bounded_model(graph_o32)
    assert len(model.boundings) == 3
    model.load_state_dict(split_prefix(golden_cfg1_gt, "sd."))
    with torch.no_grad():
        y = model.eval()(golden_cfg1_gt["x"])
    torch.testing.assert_close(y, load_npz("bounding_gt.npz")["y"], atol=5e-4, rtol=5e-4)


def test_interface_with_bounding_list_host_wiring(graph_o32, golden_interface, monkeypatch):
    """Boundings see the NORMALISED output, the de-normalisation follows -- on the first call (generic route) and on the
    later ones (normaliser folded into the first / last kernel, bounded columns finished by anemoi_bound_output)."""
    _cpu_ops.install(monkeypatch)
    iface = bounded_interface(graph_o32, golden_interface)
    iface.load_state_dict(split_prefix(golden_interface, "sd."))
    iface.eval()
    want = load_npz("bounding_gt.npz")["y_interface"]
    for _ in range(2):
        torch.testing.assert_close(iface.predict_step(golden_interface["batch"]), want, atol=1e-3, rtol=1e-3)


# ------------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("case", range(5))
def test_reference_exact_value_cases_on_the_kernel(case):
    from anemoi_models_amd import ops

    boundings, want = reference_cases()[case]
    got = ops.bound_output(input_tensor().cuda(), *op_tensors(compile_boundings(boundings), "cuda"))
    assert torch.equal(got.cpu(), want)
    x = input_tensor().cuda()
    for b in boundings:  # the module route on device tensors
        x = b(x)
    assert torch.equal(x.cpu(), want)


@pytest.mark.gpu
def test_bound_output_kernel_vs_module_chain_large():
    from anemoi_models_amd import ops

    n2i = {f"v{i}": i for i in range(80)}
    chain = [
        ReluBounding(variables=["v5", "v0", "v79"], name_to_index=n2i),
        FractionBounding(variables=["v1", "v2", "v3"], name_to_index=n2i, min_val=0.0, max_val=1.0, total_var="v2"),
        HardtanhBounding(variables=["v0", "v40"], name_to_index=n2i, min_val=-0.25, max_val=0.5),
    ]
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 1, 70001, 80, generator=g) * 2
    x[0, 0, 0, 1] = float("nan")
    want = x.clone()
    for b in chain:
        want = b(want)
    op_list = compile_boundings(chain)
    cols = bounded_columns(op_list)
    mul = torch.rand(len(cols), generator=g) + 0.5
    add = torch.randn(len(cols), generator=g)
    want[..., cols] = (want[..., cols] - add) / mul
    got = ops.bound_output(x.cuda(), *op_tensors(op_list, "cuda"),
                           fin=(torch.tensor(cols, dtype=torch.int32).cuda(), mul.cuda(), add.cuda()))
    torch.testing.assert_close(got.cpu(), want, atol=0, rtol=1e-6, equal_nan=True)


@pytest.mark.gpu
def test_model_with_bounding_list_vs_reference_golden(graph_o32, golden_cfg1_gt):
    model = bounded_model(graph_o32)
    model.load_state_dict(split_prefix(golden_cfg1_gt, "sd."))
    model = model.cuda().eval()
    with torch.no_grad():
        y = model(golden_cfg1_gt["x"].cuda()).cpu()
    want = load_npz("bounding_gt.npz")["y"]
    assert float((y - want).abs().max() / want.abs().max()) < 1e-4


@pytest.mark.gpu
def test_interface_with_bounding_list_vs_reference_golden(graph_o32, golden_interface):
    iface = bounded_interface(graph_o32, golden_interface)
    iface.load_state_dict(split_prefix(golden_interface, "sd."))
    iface = iface.cuda().eval()
    want = load_npz("bounding_gt.npz")["y_interface"]
    batch = golden_interface["batch"].cuda()
    for _ in range(2):  # generic route, then the normaliser-fused route with anemoi_bound_output finishing the columns
        y = iface.predict_step(batch).cpu()
        assert float((y - want).abs().max() / want.abs().max()) < 1e-4
    assert iface._normalizer_affines(batch) is not None
