"""pytest configuration: markers + shared fixtures."""

import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_npz(name):
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: torch.from_numpy(z[k]) for k in z.files}


def split_prefix(d, prefix):
    """Sub-dictionary of keys starting with ``prefix`` (prefix stripped)."""
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


@pytest.fixture(scope="session")
def golden_blocks():
    return load_npz("blocks.npz")


@pytest.fixture(scope="session")
def golden_index_ops():
    return load_npz("index_ops.npz")


@pytest.fixture(scope="session")
def golden_cfg1_gt():
    return load_npz("cfg1_gt.npz")


@pytest.fixture(scope="session")
def golden_interface():
    return load_npz("interface_gt.npz")


@pytest.fixture(scope="session")
def golden_hier_gt():
    return load_npz("hier_gt.npz")


@pytest.fixture(scope="session")
def graph_hier():
    from anemoi_models_amd.graphs.synthetic import build_hierarchical_graph

    return build_hierarchical_graph("o32", (2, 1))


@pytest.fixture(scope="session")
def golden_cfg1_gnn():
    return load_npz("cfg1_gnn.npz")


@pytest.fixture(scope="session")
def golden_cfg1_gnn_all():
    return load_npz("cfg1_gnn_all.npz")


@pytest.fixture(scope="session")
def golden_cfg1_tfm():
    return load_npz("cfg1_tfm.npz")


@pytest.fixture(scope="session")
def graph_o32():
    from anemoi_models_amd.graphs.synthetic import build_graph

    return build_graph("o32_ico2")
