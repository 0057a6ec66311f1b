"""graph files (``torch.save``d ``HeteroData`` as anemoi-graphs writes them) -> the model, and the
on-disk plan cache keyed on the content of the graph."""

import os
import sys
import types

import pytest
import torch

import _cpu_ops
from conftest import split_prefix
from anemoi_models_amd import runtime
from anemoi_models_amd.graphs import io as graph_io
from test_host_logic import build_model


def _same_graph(a, b):
    assert sorted(a.node_types) == sorted(b.node_types)
    assert sorted(a.edge_types) == sorted(b.edge_types)
    for name in a.node_types:
        assert torch.equal(a[name].x, b[name].x) and a[name].num_nodes == b[name].num_nodes
    for key in a.edge_types:
        for k, v in a[key].items():
            assert torch.equal(v, b[key][k]), (key, k)


def test_save_load_round_trip_and_model_on_the_loaded_graph(graph_o32, golden_cfg1_gt, tmp_path, monkeypatch):
    path = str(tmp_path / "graph.pt")
    graph_io.save_graph(graph_o32, path)
    loaded = graph_io.load_graph(path)
    _same_graph(graph_o32, loaded)
    assert runtime.graph_hash(loaded) == runtime.graph_hash(graph_o32)
    _cpu_ops.install(monkeypatch)
    model = build_model(loaded)
    model.load_state_dict(split_prefix(golden_cfg1_gt, "sd."))
    with torch.no_grad():
        y = model.eval()(golden_cfg1_gt["x"])
    torch.testing.assert_close(y, golden_cfg1_gt["y"], atol=5e-4, rtol=5e-4)


def _fake_pyg_modules():
    """Classes with torch-geometric's module paths and pickle layout (PyG 2.3 / 2.4: ``HeteroData.__dict__`` holds
    ``_global_store`` / ``_node_store_dict`` / ``_edge_store_dict``, a storage keeps its attributes in ``_mapping``)."""
    mods = {}
    for name in ("torch_geometric", "torch_geometric.data", "torch_geometric.data.hetero_data",
                 "torch_geometric.data.storage"):
        mods[name] = types.ModuleType(name)

    def make(module, cls_name):
        cls = type(cls_name, (), {"__module__": module})
        setattr(mods[module], cls_name, cls)
        return cls

    storages = {n: make("torch_geometric.data.storage", n) for n in ("BaseStorage", "NodeStorage", "EdgeStorage")}
    hetero = make("torch_geometric.data.hetero_data", "HeteroData")
    return mods, hetero, storages


def test_load_a_heterodata_pickle_without_torch_geometric(graph_o32, tmp_path):
    """A file with the pickle layout of ``torch.save(HeteroData)`` loads into a ``GraphData`` when torch-geometric is not
    installed (the layout is restated from the PyG source: see graphs/io.py)."""
    pytest.importorskip("torch")
    if "torch_geometric" in sys.modules and not isinstance(sys.modules["torch_geometric"], types.ModuleType):
        pytest.skip("a real torch_geometric is installed")
    mods, hetero, st = _fake_pyg_modules()
    sys.modules.update(mods)
    try:
        obj = hetero()
        glob = st["BaseStorage"]()
        glob.__dict__.update({"_mapping": {}, "_parent": obj})
        nodes, edges = {}, {}
        for name, store in graph_o32.node_items():
            s = st["NodeStorage"]()
            s.__dict__.update({"_mapping": dict(store), "_key": name, "_parent": obj})
            nodes[name] = s
        for key in graph_o32.edge_types:
            s = st["EdgeStorage"]()
            s.__dict__.update({"_mapping": dict(graph_o32[key]), "_key": key, "_parent": obj})
            edges[key] = s
        obj.__dict__.update({"_global_store": glob, "_node_store_dict": nodes, "_edge_store_dict": edges})
        path = str(tmp_path / "hetero.pt")
        torch.save(obj, path)
    finally:
        for name in mods:
            sys.modules.pop(name, None)
    with pytest.raises(Exception):  # the stock unpickler cannot resolve torch_geometric.* here
        torch.load(path, weights_only=False)
    loaded = graph_io.load_graph(path)
    _same_graph(graph_o32, loaded)
    assert runtime.graph_hash(loaded) == runtime.graph_hash(graph_o32)
    with pytest.raises(ValueError):
        torch.save({"not": "a graph"}, path)
        graph_io.load_graph(path)


def test_graph_hash_is_content_sensitive(graph_o32):
    h0 = runtime.graph_hash(graph_o32)
    g2 = graph_o32.to("cpu")  # a copy of the stores
    key = ("hidden", "to", "hidden")
    ei = g2[key]["edge_index"].clone()
    ei[0, 0] = (ei[0, 0] + 1) % g2["hidden"].num_nodes
    g2[key]["edge_index"] = ei
    assert runtime.graph_hash(g2) != h0
    assert runtime.graph_hash(graph_o32) == h0


def test_plan_cache_on_disk(tmp_path, monkeypatch):
    g = torch.Generator().manual_seed(0)
    ei = torch.stack([torch.randint(0, 50, (400,), generator=g), torch.randint(0, 30, (400,), generator=g)])
    relabel = torch.randperm(30, generator=g)
    runtime.set_plan_cache_dir(str(tmp_path))
    try:
        p1 = runtime.PlanCache().get(ei, 50, 30, dst_map=relabel)
        files = sorted(os.listdir(tmp_path))
        assert len(files) == 1 and files[0].startswith("edgeplan-") and files[0].endswith(".pt")
        # a new process / module instance: the plan comes from the file, nothing is sorted again
        calls = []
        real_build = runtime.build_edge_plan
        monkeypatch.setattr(runtime, "build_edge_plan", lambda *a, **k: calls.append(1) or real_build(*a, **k))
        p2 = runtime.PlanCache().get(ei.clone(), 50, 30, dst_map=relabel.clone())  # other tensors, same content
        assert calls == []
        for a, b in ((p1.rowptr, p2.rowptr), (p1.col, p2.col), (p1.perm, p2.perm)):
            assert torch.equal(a, b) and a.dtype == b.dtype
        assert (p2.n_src, p2.n_dst) == (50, 30)
        # different content -> different key; batched plan -> different key
        runtime.PlanCache().get(ei.flip(1).contiguous(), 50, 30, dst_map=relabel)
        inc = torch.tensor([[50], [30]])
        runtime.PlanCache().get(ei, 100, 60, 2, inc)
        assert len(calls) == 2 and len(os.listdir(tmp_path)) == 3
        # a truncated file is ignored and replaced
        path = os.path.join(tmp_path, files[0])
        with open(path, "wb") as f:
            f.write(b"\x00" * 10)
        p3 = runtime.PlanCache().get(ei, 50, 30, dst_map=relabel)
        assert len(calls) == 3 and torch.equal(p3.col, p1.col)
        assert runtime.load_edge_plan(path, "cpu") is not None
        # a stale / corrupt but well-formed file must not become out-of-bounds gathers: column beyond the source set,
        # perm that is no permutation, row pointers that go backwards, a plan for another node count -> rebuilt
        good = torch.load(path, weights_only=True)
        for field, bad in (("col", lambda t: t.index_fill(0, torch.tensor([3]), 50)),
                           ("col", lambda t: t.index_fill(0, torch.tensor([0]), -1)),
                           ("perm", lambda t: t.index_fill(0, torch.tensor([5]), int(t[6]))),
                           ("rowptr", lambda t: torch.cat([t[:2], t[1:2] - 1, t[3:]]))):
            d = dict(good)
            d[field] = bad(good[field].clone())
            torch.save(d, path)
            assert runtime.load_edge_plan(path, "cpu", 50, 30) is None, field
        torch.save(good, path)
        assert runtime.load_edge_plan(path, "cpu", 50, 30) is not None
        assert runtime.load_edge_plan(path, "cpu", 51, 30) is None and runtime.load_edge_plan(path, "cpu", 50, 31) is None
    finally:
        runtime.set_plan_cache_dir(None)
    n = len(os.listdir(tmp_path))
    runtime.PlanCache().get(ei[:, :10].contiguous(), 50, 30)  # cache off: nothing written
    assert len(os.listdir(tmp_path)) == n


@pytest.mark.gpu
def test_model_from_graph_file_with_disk_plan_cache_on_the_gpu(graph_o32, golden_cfg1_gt, tmp_path):
    """Graph file -> model on the device with the plan cache directory set: first model writes the plans, a second model
    (fresh caches) reads them; both reproduce the reference output."""
    path = str(tmp_path / "graph.pt")
    graph_io.save_graph(graph_o32, path)
    runtime.set_plan_cache_dir(str(tmp_path / "plans"))
    try:
        outs = []
        for _ in range(2):
            model = build_model(graph_io.load_graph(path))
            model.load_state_dict(split_prefix(golden_cfg1_gt, "sd."))
            model = model.cuda().eval()
            with torch.no_grad():
                outs.append(model(golden_cfg1_gt["x"].cuda()).cpu())
            n_files = len(os.listdir(tmp_path / "plans"))
            assert n_files == 3  # encoder, processor, decoder
        want = golden_cfg1_gt["y"]
        for y in outs:
            assert float((y - want).abs().max() / want.abs().max()) < 1e-4
        assert torch.equal(outs[0], outs[1])
    finally:
        runtime.set_plan_cache_dir(None)
