#!/usr/bin/env python
"""Headline benchmark: mesh-node updates/sec of one full encoder -> processor -> decoder forward step.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json / SURVEY.md section 8d): ``N_mesh x num_processor_blocks / t_fwd`` with ``t_fwd`` the wall time
of one ``AnemoiModelEncProcDec.forward`` (batch 1, eval, no grad), inputs resident in HBM.  Default workload: BASELINE
config 3 (N320 -> ico-6, 16 GraphTransformer blocks, 1024 channels, 2 x 90 input variables) on synthetic graph /
state / weights, bf16 storage with f32 accumulation.  One process per GPU; with N > 1 the mesh is node-partitioned
over the ranks of one model group (strong scaling: the same single forward step is shared by all ranks).

The JSON line also carries
  roofline      the dominant kernel (fused Linear, MFMA bound): algorithmic flops / measured kernel time, live,
                from HIP events on the launch stream in a separate instrumented pass (not inside the timed region);
  roofline_edge the fused gather/scatter edge kernel against the HBM roofline (algorithmic bytes of section 8d);
  cpu_baseline  the CPU oracle (plain PyTorch, same algorithm as the reference) timed on this host's cores on the same
                forward (all of it by default; --cpu-blocks N runs N of the identical processor blocks and scales), with
                the parity of this run's device results against the oracle's outputs beside it.
``value`` / ``ms_per_step`` come from the K-step bracket the driver's contract prescribes (mean); ``ms_per_step_median``
/ ``value_at_median`` are the per-step median (device events on the launch stream, rank 0) SURVEY section 8d asks for.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}  # dense MFMA peaks (bf16 / exact-f32 MFMA)
# parity bounds of the run itself (max |a - b| / max |b| of the prediction): north_star's 1e-3 for f32; bf16 storage is
# reported against the f32 oracle, measured 2.9e-3 at config 3 -- beyond 1e-2 the line is an error, not a result
PARITY_BOUND = {"bf16": 1e-2, "fp32": 1e-3}
# the encoder output (mesh latent, 1024 channels, BEFORE the processor) is gated as well: bf16 measured 9.8e-3 at config 3
# -- the largest entries of a 1024-channel latent carry one bf16 rounding of the residual stream (attribution:
# profiles/r05_latent_error.txt) --, f32 2e-6
LATENT_BOUND = {"bf16": 2e-2, "fp32": 1e-3}
# test hook (tests/test_bench_contract.py): scales every parity bound, so that the "a failed comparison prints ONE error line
# and no result line" contract can be exercised on a healthy build
_BOUND_SCALE = float(os.environ.get("ANEMOI_AMD_BENCH_PARITY_BOUND_SCALE", "1"))
if not 0.0 < _BOUND_SCALE <= 1.0:  # the hook can only TIGHTEN the gates: a relaxed gate must never print a normal line
    raise SystemExit(f"ANEMOI_AMD_BENCH_PARITY_BOUND_SCALE={_BOUND_SCALE}: only factors in (0, 1] are accepted")
if _BOUND_SCALE != 1.0:
    PARITY_BOUND = {k: v * _BOUND_SCALE for k, v in PARITY_BOUND.items()}
    LATENT_BOUND = {k: v * _BOUND_SCALE for k, v in LATENT_BOUND.items()}

def host_threads() -> int:
    """Threads the CPU oracle may use: the affinity mask, cut by a cgroup CPU quota when the box has one (the GPU boxes of
    this pool show 256 logical CPUs and a quota of 16: 128 torch threads on them are slower than 16, and "cores": 128 would
    misstate what the baseline ran on)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


WORKLOADS = {
    # name: (graph, channels, processor blocks, heads, description)
    "cfg1": ("o32_ico2", 64, 4, 16, "O32->ico-2, 4 GT blocks, 64 ch"),
    "cfg2": ("o96_ico5", 512, 16, 16, "O96->ico-5, 16 GT blocks, 512 ch"),
    "cfg3": ("n320_ico6", 1024, 16, 16, "N320->ico-6, 16 GT blocks, 1024 ch"),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--processor", default="GraphTransformer", choices=["GraphTransformer", "GNN", "Transformer"],
                    help="processor family (BASELINE config 5 = --workload cfg2 --processor GNN)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-blocks", type=int, default=None,
                    help="processor blocks the CPU-baseline sample runs (default: all of them -- at config 3 the whole "
                         "forward takes ~2.5 min on the GPU box's 128 host threads and doubles as the full-size parity check; "
                         "fewer blocks: the sample is scaled); the encoder and decoder always run in full")
    ap.add_argument("--rollout", type=int, default=1,
                    help="autoregressive forecasts per step (BASELINE config 4 = --rollout 4): forward, then the "
                         "in-place input update anemoi_advance_input, repeated")
    ap.add_argument("--detail", action="store_true", help="print a per-shape kernel table to stderr")
    ap.add_argument("--hipgraph", action="store_true",
                    help="replay the forward as one captured HIP graph (single GPU; pays off on the small workloads)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the 'secondary' block of the default line (Transformer processor @ config 3 with the mesh "
                         "attention's MFMA roofline, config 2, config 5: 5 steps each, behind the timed region)")
    return ap.parse_args()


def build(workload: str, device, processor: str = "GraphTransformer", graph=None):
    from anemoi_models_amd.graphs.synthetic import build_graph
    from anemoi_models_amd.models import AnemoiModelEncProcDec
    from anemoi_models_amd.utils.indices import SimpleDataIndices
    from anemoi_models_amd.utils.presets import model_config

    graph_name, channels, layers, heads, _ = WORKLOADS[workload]
    if graph is None:
        graph = build_graph(graph_name)
    idx = SimpleDataIndices(n_prognostic=80, n_forcing=10, n_diagnostic=0)
    torch.manual_seed(1234)
    with torch.device(device):  # random-init the weights directly in HBM
        model = AnemoiModelEncProcDec(model_config=model_config(processor, channels, layers, heads),
                                      data_indices=idx, graph_data=graph.to(device))
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("trainable"):
                p.normal_(0.0, 0.1)
    model = model.to(device).eval()
    x = torch.randn((1, 2, 1, graph["data"].num_nodes, idx.num_input), generator=torch.Generator().manual_seed(7))
    return model, graph, x.to(device), idx


def detail_table(records, dtype_name: str) -> None:
    from collections import defaultdict

    agg = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    for name, start, end, work in records:
        key = (name,) + tuple(sorted((k, v) for k, v in work.items() if k not in ("flops", "bytes", "fused_bytes")))
        a = agg[key]
        a[0] += 1
        a[1] += start.elapsed_time(end)
        a[2] += work.get("flops", 0)
        a[3] += work.get("bytes", 0)
    print(f"{'kernel / shape':70s} {'n':>4s} {'avg_ms':>9s} {'TFLOP/s':>9s} {'GB/s':>9s}", file=sys.stderr)
    for key, (n, ms, fl, by) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        label = key[0] + " " + " ".join(f"{k}={v}" for k, v in key[1:])
        print(f"{label:70s} {n:4d} {ms / n:9.4f} {fl / ms / 1e9 if fl else 0:9.1f} {by / ms / 1e6 if by else 0:9.1f}",
              file=sys.stderr)


def reference_linear_flops(model) -> float:
    """ALGORITHMIC flops of the nn.Linear layers one forward of the reference runs (per
    GraphTransformer block 26 N C^2 + the lin_edge rows; mappers: embeddings, k | v on the sources, self | q on the
    destinations, lin_edge, projection, node MLP, extraction), batch 1: 2 x rows x in x out per layer, whatever this
    package folds away or adds (embedding fold, lin_edge fold).  The numerator of ``roofline.achieved``; 0 for processor
    families it does not cover (then the executed flops are reported and labelled)."""
    from anemoi_models_amd.layers.mapper import GraphTransformerBaseMapper
    from anemoi_models_amd.layers.processor import GraphTransformerProcessor

    if not (isinstance(model.encoder, GraphTransformerBaseMapper) and isinstance(model.decoder, GraphTransformerBaseMapper)
            and isinstance(model.processor, GraphTransformerProcessor)):
        return 0.0

    def lin(layer, rows):
        return 2.0 * rows * layer.in_features * layer.out_features

    def mapper(m, n_src, n_dst):
        blk, e = m.proc, m.edge_attr.shape[0]
        f = lin(blk.lin_key, n_src) + lin(blk.lin_value, n_src) + lin(blk.lin_query, n_dst) + lin(blk.lin_self, n_dst)
        f += lin(blk.lin_edge, e) + lin(blk.projection, n_dst) + lin(blk.node_dst_mlp[1], n_dst) + lin(blk.node_dst_mlp[3], n_dst)
        f += lin(m.emb_nodes_dst, n_dst)
        if hasattr(m, "emb_nodes_src"):
            f += lin(m.emb_nodes_src, n_src)
        if hasattr(m, "node_data_extractor"):
            f += lin(m.node_data_extractor[1], n_dst)
        return f

    na = model.node_attributes
    n_g, n_m = na.num_nodes[model._graph_name_data], na.num_nodes[model._graph_name_hidden]
    total = mapper(model.encoder, n_g, n_m) + mapper(model.decoder, n_m, n_g)
    e_proc = model.processor.edge_attr.shape[0]
    for chunk in model.processor.proc:
        for blk in chunk.blocks:
            total += sum(lin(l, n_m) for l in (blk.lin_key, blk.lin_value, blk.lin_query, blk.lin_self, blk.projection,
                                                blk.node_dst_mlp[1], blk.node_dst_mlp[3])) + lin(blk.lin_edge, e_proc)
    return total


_SLEEP_TICKS_PER_MS = None


def hold_stream(ms: float) -> None:
    """Park the launch stream for about ``ms`` milliseconds (a spin kernel), so that what the host enqueues next is queued
    AHEAD of the GPU.  The instrumented forward records two events per launch; on the small configurations (config 2: ~200
    launches of 10 - 40 us) the host cannot keep up with that, the stream runs dry between launches and every event pair
    then brackets the kernel PLUS the wait for its submission -- 3.0 - 4.5 ms of "Linear time" inside a 2.96 ms step, box
    by box, where rocprofv3 measures 2.61.  Behind a parked stream the events bracket back-to-back kernels, as the
    profiler's timestamps do (2.71 ms)."""
    global _SLEEP_TICKS_PER_MS
    sleep = getattr(torch.cuda, "_sleep", None)
    if sleep is None:
        return
    if _SLEEP_TICKS_PER_MS is None:  # what one tick of the spin kernel's clock is worth on this box
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        sleep(1_000_000)
        e1.record()
        torch.cuda.synchronize()
        _SLEEP_TICKS_PER_MS = 1_000_000 / max(e0.elapsed_time(e1), 1e-3)
    sleep(int(ms * _SLEEP_TICKS_PER_MS))


HOLD_BELOW_MS, HOLD_MS = 10.0, 8.0  # steps shorter than HOLD_BELOW_MS: the instrumented forward is enqueued behind a hold of >= 8 ms


def profile_pass(model, x, group, dtype_name: str, detail: bool = False, traffic_ok: bool = True, step_ms: float = 1e9):
    """One instrumented forward: HIP events around every kernel launch, on the launch stream.  Where a step is short enough
    for the host's submission rate to show in the event pairs (``step_ms < HOLD_BELOW_MS``: configs 1, 2, 5, a rank of a
    partition) the forward is enqueued behind a parked stream (:func:`hold_stream`); the long steps are GPU-bound without it,
    and a GPU that idled through a hold starts them at a lower clock (config 3: Linear 31.0 ms behind a 25-ms hold, 30.3
    without, 30.4 by rocprofv3; `gpurun_out/r06_s44`)."""
    from anemoi_models_amd import ops

    def instrumented():
        ops.PROFILE = []
        t0 = time.perf_counter()
        with torch.no_grad():
            model(x, group) if group is not None else model(x)
        host_ms = (time.perf_counter() - t0) * 1e3
        torch.cuda.synchronize()
        return host_ms

    torch.cuda.synchronize()
    hold = os.environ.get("ANEMOI_AMD_BENCH_HOLD_MS")
    if hold is not None:
        hold = float(hold)
    elif step_ms < HOLD_BELOW_MS:
        # long enough for the host to enqueue the WHOLE instrumented forward: a rehearsal says how long that takes on this
        # box (4 - 9 ms at config 2, box by box: a fixed 8 ms was not always enough)
        hold = min(40.0, max(HOLD_MS, 1.5 * instrumented() + 2.0))
    else:
        hold = 0.0
    if hold > 0:
        hold_stream(hold)
    instrumented()
    records, ops.PROFILE = ops.PROFILE, None
    if detail:
        detail_table(records, dtype_name)
    agg = {}
    for name, start, end, work in records:
        a = agg.setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "fused_bytes": 0.0})
        a["launches"] += 1
        a["ms"] += start.elapsed_time(end)
        a["flops"] += work.get("flops", 0)
        a["bytes"] += work.get("bytes", 0)
        a["fused_bytes"] += work.get("fused_bytes", 0)
    out = {}
    if "linear" in agg and agg["linear"]["ms"] > 0:
        a = agg["linear"]
        # numerator = the ALGORITHMIC flops of the reference's Linear layers for this forward, not the flops this
        # package happens to execute (folds remove some products and add others); the executed figure is given beside it
        # (N > 1: this rank's share of them -- the partition splits the rows, so 1 / world of every Linear is this rank's
        # useful work; what the rank executes on top of that, e.g. k | v of halo rows, is not credited)
        world = 1 if group is None else group.size()
        algo = reference_linear_flops(model) / world
        flops = algo if algo > 0 else a["flops"]
        achieved = flops / (a["ms"] * 1e-3) / 1e12
        peak = MFMA_PEAK_TFLOPS[dtype_name]
        out["roofline"] = {
            "kernel": "anemoi::linear_bf16_w4_kernel / linear_kernel (anemoi_linear: fused Linear, MFMA)", "bound": "mfma", "achieved": round(achieved, 2),
            "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": None,
            "launches": a["launches"], "avg_launch_ms": round(a["ms"] / a["launches"], 4),
            "flops_per_launch": flops / a["launches"], "bytes_per_launch": a["bytes"] / a["launches"],
            "flops_model": ("algorithmic: 2 x rows x in x out of every nn.Linear of the reference forward (SURVEY 8d)"
                            + (f", 1/{world} of them: this rank's rows" if world > 1 else "")
                            if algo > 0 else "executed by this package's launches"),
            # the same kernel time against the flops the launches really execute (folds remove some products, add others)
            "frac_executed": round(a["flops"] / (a["ms"] * 1e-3) / 1e12 / peak, 4),
            "executed_flops_per_launch": a["flops"] / a["launches"],
            "executed_tflops": round(a["flops"] / (a["ms"] * 1e-3) / 1e12, 2),
            "share_of_step": None,
        }
    if "gt_edge_attention" in agg and agg["gt_edge_attention"]["ms"] > 0:
        a = agg["gt_edge_attention"]
        achieved = a["bytes"] / (a["ms"] * 1e-3) / 1e9
        out["roofline_edge"] = {
            "kernel": "anemoi::gt_edge_attention_folded_{sched,runs,}_kernel (fused gather / score / segment softmax / scatter)", "bound": "hbm",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "launches": a["launches"],
            "avg_launch_ms": round(a["ms"] / a["launches"], 4), "bytes_per_launch": a["bytes"] / a["launches"],
            # the same launches against the bytes INCLUDING the operands of the kernel's fusions (x_r read, u read, t written:
            # not in SURVEY 8d's figure); beside frac, not instead of it
            "frac_fused_operands": round(a["fused_bytes"] / (a["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "fused_operand_bytes_per_launch": a["fused_bytes"] / a["launches"],
        }
    if "mhsa" in agg and agg["mhsa"]["ms"] > 0:  # Transformer processor: mesh-node self attention (MFMA-bound)
        a = agg["mhsa"]
        achieved = a["flops"] / (a["ms"] * 1e-3) / 1e12
        peak = MFMA_PEAK_TFLOPS[dtype_name]
        out["roofline_mhsa"] = {
            "kernel": "anemoi::mhsa_bf16_w4_kernel (D = 64, global) / mhsa_bf16_kernel<D> (anemoi_mhsa: flash attention on v_mfma_f32_32x32x16_bf16)",
            "bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": None, "launches": a["launches"],
            "avg_launch_ms": round(a["ms"] / a["launches"], 4), "flops_per_launch": a["flops"] / a["launches"],
        }
    for name in ("gather_add_act", "segment_sum"):  # GNN processor: HBM-bound edge kernels
        if name in agg and agg[name]["ms"] > 0:
            a = agg[name]
            achieved = a["bytes"] / (a["ms"] * 1e-3) / 1e9
            out["roofline_" + name] = {
                "kernel": "anemoi::" + name + "_kernel", "bound": "hbm", "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                "launches": a["launches"], "avg_launch_ms": round(a["ms"] / a["launches"], 4),
                "bytes_per_launch": a["bytes"] / a["launches"],
            }
    # HBM traffic per launch: PMC counters cannot be read from inside the process, so the figure is the one of the latest
    # COMMITTED rocprofv3 --pmc passes of this same command (tools/refresh_profiles.sh -> profiles/rNN_traffic.json) and
    # is labelled as such ("traffic_source": not measured in this run); null when no pass exists for this configuration.
    try:
        if not traffic_ok or group is not None:  # the PMC passes are taken on config 3 / bf16 / one GPU only
            raise KeyError("no PMC pass for this run")
        import glob

        latest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_traffic.json")))[-1]
        with open(latest) as f:
            traffic = json.load(f)["kernels"]
        label = f"profiles/{os.path.basename(latest)}: rocprofv3 --pmc FETCH_SIZE (x2) + WRITE_SIZE passes of this command, " \
                "committed earlier -- NOT measured in this run"
        if "roofline" in out and dtype_name == "bf16":
            out["roofline"]["traffic"] = traffic["linear"]["traffic_bytes_per_launch"]
            out["roofline"]["traffic_source"] = label
        if "roofline_edge" in out and dtype_name == "bf16":
            out["roofline_edge"]["traffic"] = traffic["gt_edge_attention"]["traffic_bytes_per_launch"]
            out["roofline_edge"]["traffic_source"] = label
    except (OSError, KeyError, ValueError, IndexError):
        pass
    total_ms = sum(a["ms"] for a in agg.values())
    out["kernel_time_ms"] = {k: round(v["ms"], 3) for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}
    if "roofline" in out and total_ms > 0:
        out["roofline"]["share_of_step"] = round(agg["linear"]["ms"] / total_ms, 3)
    return out


def _rel(a, b) -> float:
    return float((a.float().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _per_variable_rel(a, b):
    """SURVEY 8d parity gate: per output variable ``||a - b||_inf / ||b||_inf``."""
    a, b = a.float().cpu().flatten(0, -2), b.flatten(0, -2)
    return ((a - b).abs().max(dim=0).values / b.abs().max(dim=0).values.clamp_min(1e-30)).tolist()


def device_forward_with_latent(model, x):
    """One forward on the device that also hands back the encoder output (mesh latent) in the EXTERNAL node order."""
    captured = {}
    native = model.encoder.native

    def capture(*a, **k):
        out = native(*a, **k)
        captured["latent"] = out[1] if isinstance(out, tuple) else out
        return out

    model.encoder.native = capture
    try:
        with torch.no_grad():
            y = model(x)
    finally:
        del model.encoder.native  # back to the class's method
    _, inv = model._mesh_order(x.device)
    return y, captured["latent"][: inv.numel()].index_select(0, inv)[:, : model.num_channels]


def f32_leg(model, x, steps: int = 5):
    """The north star's own parity statement (1e-3 rel fp32) needs an f32 forward of the SAME model / input: the exact-f32
    MFMA route of this package (``ANEMOI_AMD_DTYPE=fp32``), timed over ``steps`` forwards after one warm-up."""
    before = os.environ.get("ANEMOI_AMD_DTYPE")
    os.environ["ANEMOI_AMD_DTYPE"] = "fp32"
    try:
        y, latent = device_forward_with_latent(model, x)
        y, latent = y.float().cpu(), latent.float().cpu()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            for _ in range(steps):
                model(x)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
    finally:
        if before is None:
            del os.environ["ANEMOI_AMD_DTYPE"]
        else:
            os.environ["ANEMOI_AMD_DTYPE"] = before
    torch.cuda.empty_cache()
    return y, latent, ms, steps


def cpu_baseline(model, graph, x, idx, n_blocks: int, hip_latent=None, hip_y=None, f32=None):
    """CPU oracle (plain-PyTorch restatement of the reference algorithm, oracle/reference_path.py) timed on this host's
    cores on the SAME forward as the headline value: input assembly, encoder, processor, decoder, prognostic residual.

    Bounded sample: the encoder and the decoder run in full; of the ``L`` identical processor blocks the first
    ``n_blocks`` run (all of them when ``n_blocks >= L``) and the step time is ``t_enc + t_blocks * L / n_blocks + t_dec``
    -- the blocks are the same code on the same shapes, so the scaling is exact up to cache warmth.  Mapper blocks use
    the reference's own inference chunking (``num_chunks`` 8 at the N320 grid) to bound host memory.
    """
    import platform

    from oracle import reference_path as ref  # checker / baseline only

    if "OMP_NUM_THREADS" not in os.environ:
        torch.set_num_threads(host_threads())

    sd = {k: (v.detach().float() if v.is_floating_point() else v.detach()).cpu() for k, v in model.state_dict().items()}
    data, hidden = model._graph_name_data, model._graph_name_hidden

    def edges(mod):
        return mod.edge_attr.detach().float().cpu(), mod.edge_index_base.cpu()

    heads = model.processor.proc[0].blocks[0].num_heads
    n_layers = sum(len(chunk.blocks) for chunk in model.processor.proc)
    names = [f"processor.proc.{ci}.blocks.{bi}" for ci, chunk in enumerate(model.processor.proc)
             for bi in range(len(chunk.blocks))]
    n_blocks = max(1, min(n_blocks, n_layers))
    n_grid, n_mesh = graph[data].num_nodes, graph[hidden].num_nodes
    mapper_chunks = 8 if n_grid > 200_000 else 1
    xc = x.detach().float().cpu()
    b, t, ens, g, v = xc.shape
    tick = time.perf_counter
    with torch.no_grad():
        t0 = tick()
        x_data = torch.cat((xc.permute(0, 2, 3, 1, 4).reshape(b * ens * g, t * v), ref.node_attributes(sd, data, b)), dim=-1)
        x_hidden = ref.node_attributes(sd, hidden, b)
        ea, ei = edges(model.encoder)
        _, x_latent = ref.gt_forward_mapper(sd, "encoder", x_data, x_hidden, ea, ei, b, heads, "GELU", mapper_chunks)
        t_enc = tick() - t0
        ea, ei = edges(model.processor)
        edge_attr = ref.trainable_tensor(ea, sd.get("processor.trainable.trainable"), b)
        t0 = tick()
        xm = x_latent
        for name in names[:n_blocks]:
            xm = ref.gt_processor_block(sd, name, xm, edge_attr, ei, heads)
        t_blk = tick() - t0
        ea, ei = edges(model.decoder)
        t0 = tick()
        y = ref.gt_backward_mapper(sd, "decoder", xm + x_latent, x_data, ea, ei, b, heads, "GELU", mapper_chunks)
        y = y.reshape(b, ens, g, -1).clone()
        pin, pout = [int(i) for i in model._internal_input_idx], [int(i) for i in model._internal_output_idx]
        y[..., pout] += xc[:, -1, :, :, pin]
        t_dec = tick() - t0
    t_fwd = t_enc + t_blk * n_layers / n_blocks + t_dec
    # parity of THIS run's device results against the oracle outputs the baseline has just produced on the same input
    # (the oracle is the checker here, never the product): the encoder output (mesh latent, 1024 ch) always; the final
    # prediction when every processor block was run
    rel = _rel
    parity, parity_fp32 = {}, None
    if hip_latent is not None:
        parity["encoder_out_rel_err"] = rel(hip_latent, x_latent)
    if hip_y is not None and n_blocks == n_layers:
        parity["output_rel_err"] = rel(hip_y, y)
        pv = _per_variable_rel(hip_y, y)
        parity["per_variable_rel_err_max"] = max(pv)
    if parity:
        parity["vs"] = "CPU oracle (f32) on the same weights / input, max |a - b| / max |b|"
    if f32 is not None:  # the exact-f32 route of the same model on the same input: north_star's "within 1e-3 rel fp32"
        y32, latent32, ms32, steps32 = f32
        parity_fp32 = {"encoder_out_rel_err": rel(latent32, x_latent), "ms_per_step_f32": round(ms32, 3), "steps": steps32,
                       "bound": PARITY_BOUND["fp32"], "processor_blocks_compared": n_blocks,
                       "vs": "CPU oracle (f32), the device on its exact-f32 MFMA route (ANEMOI_AMD_DTYPE=fp32), same "
                             "weights / input; per variable: ||a - b||_inf / ||b||_inf of every output column"}
        if n_blocks == n_layers:
            pv = _per_variable_rel(y32, y)
            parity_fp32["output_rel_err"] = rel(y32, y)
            parity_fp32["per_variable_rel_err_max"] = max(pv)
            parity_fp32["per_variable_rel_err"] = [float(f"{v:.3e}") for v in pv]
    cpu = platform.processor() or platform.machine()
    try:
        with open("/proc/cpuinfo") as f:
            cpu = next(line.split(":", 1)[1].strip() for line in f if line.startswith("model name"))
    except (OSError, StopIteration):
        pass
    scaled = "" if n_blocks == n_layers else f" x {n_layers}/{n_blocks}"
    return {
        "value": round(n_mesh * n_layers / t_fwd, 1), "unit": "mesh-node updates/s", "cores": torch.get_num_threads(),
        "kind": "port", "cpu": cpu,
        "sample": f"the same forward on the CPU oracle (plain PyTorch f32, oracle/reference_path.py): input assembly + "
                  f"encoder {t_enc:.1f} s + {n_blocks} of {n_layers} processor blocks {t_blk:.1f} s{scaled} + decoder + "
                  f"residual {t_dec:.1f} s = {t_fwd:.1f} s per step ({n_grid} grid / {n_mesh} mesh nodes, "
                  f"{model.num_channels} ch, mapper chunks {mapper_chunks}); measured {t_enc + t_blk + t_dec:.1f} s",
        **({"parity": parity} if parity else {}),
        **({"parity_fp32": parity_fp32} if parity_fp32 else {}),
    }


SECONDARY_LEGS = (
    # name, workload, processor, (steps, warm-ups): the north star's MHSA clause and BASELINE configs 2 / 5 inside the
    # driver's own line (the O96-sized legs are 3 - 7 ms per step: 20 of them, so that the figure is not launch jitter)
    ("transformer_cfg3", "cfg3", "Transformer", (5, 2)),
    ("cfg2", "cfg2", "GraphTransformer", (20, 5)),
    ("cfg5_gnn", "cfg2", "GNN", (20, 5)),
)


def secondary_leg(workload: str, processor: str, device, dtype_name: str, graph=None, steps: int = 5, warmup: int = 2):
    """One secondary workload of the default line: ``warmup`` + ``steps`` forwards bracketed by device syncs, then one
    instrumented forward for its roofline objects (same definitions as the headline's).  Reference for the Transformer leg:
    layers/attention.py:67-112 (mesh-node MultiHeadSelfAttention), 4 * S^2 * C flops per layer against the bf16 MFMA peak."""
    model, graph, x, _ = build(workload, device, processor, graph)
    n_mesh, layers = graph["hidden"].num_nodes, WORKLOADS[workload][2]
    with torch.no_grad():
        for _ in range(warmup):
            y = model(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            y = model(x)
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    if not bool(torch.isfinite(y).all()):
        raise RuntimeError("non-finite output")
    extra = profile_pass(model, x, None, dtype_name, traffic_ok=False, step_ms=ms)
    keep = ("frac", "achieved", "unit", "bound", "avg_launch_ms", "launches", "frac_executed")
    out = {"workload": WORKLOADS[workload][4].replace("GT blocks", f"{processor} blocks") if processor != "GraphTransformer"
           else WORKLOADS[workload][4], "ms_per_step": round(ms, 3), "steps": steps, "warmup": warmup,
           "value": round(n_mesh * layers / (ms * 1e-3), 1), "unit": "mesh-node updates/s"}
    for key, val in extra.items():
        if key.startswith("roofline"):
            out[key] = {k: v for k, v in val.items() if k in keep}
    out["kernel_time_ms"] = dict(list(extra.get("kernel_time_ms", {}).items())[:4])
    del model, x, y
    torch.cuda.empty_cache()
    return out


def secondary_block(device, dtype_name: str, graph_cfg3=None):
    """``secondary`` of the default line.  A leg that fails reports ``{"error": ...}`` under its own name and never removes
    the headline; ``ANEMOI_AMD_BENCH_SECONDARY`` (comma-separated leg names) restricts the legs (tests)."""
    only = os.environ.get("ANEMOI_AMD_BENCH_SECONDARY")
    only = None if only is None else {n.strip() for n in only.split(",") if n.strip()}
    out = {}
    for name, workload, processor, (steps, warmup) in SECONDARY_LEGS:
        if only is not None and name not in only:
            continue
        try:
            out[name] = secondary_leg(workload, processor, device, dtype_name, graph_cfg3 if workload == "cfg3" else None,
                                      steps, warmup)
        except BaseException as exc:  # noqa: BLE001 -- a secondary leg must never cost the headline
            if isinstance(exc, (KeyboardInterrupt, SystemExit)):
                raise
            out[name] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
            torch.cuda.empty_cache()
    out["note"] = ("each leg: warm-ups, then `steps` timed forwards (device-synchronised wall time) and one instrumented forward "
                   "for its roofline objects, run AFTER the headline's timed region and its CPU baseline, the config-3 model "
                   "freed first")
    return out


class Stage:
    """Where the run is, for the error line -- and for the watchdog: a stage with a time limit (process-group init, the
    first collective, the partitioned-vs-single comparison: the places where a rank whose peer never arrives would block
    for ever) that overruns ends the process with ONE JSON error line and exit code 4.  No restart, no retry."""

    LIMITS_S = {"init_process_group": 300.0, "first all_reduce": 180.0, "parity_vs_single": 900.0,
                "exchange timing": 300.0}

    def __init__(self) -> None:
        self.name, self.since = "start", time.monotonic()
        scale = float(os.environ.get("ANEMOI_AMD_BENCH_WATCHDOG_SCALE", "1"))
        self.limits = {k: v * scale for k, v in self.LIMITS_S.items()}
        self._printed = False

    def __getitem__(self, i):  # stage[0] / stage[0] = "...": the list protocol the code below uses
        return self.name

    def __setitem__(self, i, name) -> None:
        self.name, self.since = name, time.monotonic()

    def watch(self) -> None:
        import threading

        def loop():
            while True:
                time.sleep(0.5)
                limit = self.limits.get(self.name)
                if limit is not None and time.monotonic() - self.since > limit:
                    print(json.dumps({"error": f"watchdog: stage '{self.name}' exceeded {limit:.0f} s (a peer rank missing "
                                               "or a collective that never completes)", "stage": self.name,
                                      "rank": int(os.environ.get("RANK", "0"))}), flush=True)
                    os._exit(4)

        threading.Thread(target=loop, daemon=True, name="bench-watchdog").start()


def self_launch(n: int, argv) -> int:
    """``python bench.py --gpus N`` without a launcher: BEFORE this process has touched the GPU, start the N ranks as
    children (``python -m torch.distributed.run --nproc-per-node N bench.py ...``, rendezvous on 127.0.0.1), relay rank
    0's JSON line -- exactly one line on stdout -- and return the worst exit code.  (A process that has initialised the
    GPU never re-launches or replaces itself; this one only imports torch.)"""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)  # the children's stderr passes through
    results, errors = [], []
    for ln in proc.stdout:
        if ln.startswith("{"):
            try:
                obj = json.loads(ln)
            except ValueError:
                obj = None
            if isinstance(obj, dict):
                (errors if "error" in obj else results).append(ln.rstrip("\n"))
                continue
        sys.stderr.write(ln)
    rc = proc.wait()
    if rc == 0 and results and not errors:
        print(results[-1], flush=True)
        return 0
    if errors:
        print(errors[0], flush=True)
    else:
        print(json.dumps({"error": f"torch.distributed.run exited with {rc} and no JSON line", "stage": "self-launch",
                          "rank": 0}), flush=True)
    return rc if rc != 0 else 3


def _run(stage, args) -> int:
    failure = None
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise RuntimeError(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks (one process per GPU: "
                           "python -m torch.distributed.run --nproc-per-node N bench.py --gpus N, or plain "
                           "python bench.py --gpus N, which launches them itself)")
    stage.watch()
    from anemoi_models_amd import _lib

    _lib.load()  # fail loudly if the HIP library is missing
    # debugging aid for 1-GPU boxes: ANEMOI_AMD_BENCH_SHARE_GPU=1 puts every rank on cuda:0 with host-staged gloo
    # collectives, so that the N > 1 code path (plans, halos, gather, timing protocol) can be run -- not timed -- there
    share_gpu = os.environ.get("ANEMOI_AMD_BENCH_SHARE_GPU", "0") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    group = None
    rccl_ranks = 1
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        stage[0] = "init_process_group"
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
        group = dist.group.WORLD
        # one collective before anything is built: a rank that cannot reach its peers over RCCL fails HERE, with a JSON
        # error line, not minutes later inside the first halo exchange
        stage[0] = "first all_reduce"
        probe = torch.ones(1, device="cpu" if share_gpu else device)
        dist.all_reduce(probe)
        rccl_ranks = dist.get_world_size()
        if int(probe.item()) != world or rccl_ranks != world:
            raise RuntimeError(f"process group has {rccl_ranks} ranks (all_reduce of ones: {probe.item()}), launched {world}")
    stage[0] = "build"

    os.environ["ANEMOI_AMD_DTYPE"] = args.dtype
    model, graph, x, idx = build(args.workload, device, args.processor)
    n_mesh = graph["hidden"].num_nodes
    layers = WORKLOADS[args.workload][2]

    graphed = None
    if args.hipgraph and group is None:
        from anemoi_models_amd.runtime import GraphedForward

        graphed = GraphedForward(model, x)

    cmap = None
    if args.rollout > 1:  # prognostic inputs <- their output column; forcings persist (no data source in a benchmark)
        from anemoi_models_amd import ops

        cmap = torch.full((idx.num_input,), -1, dtype=torch.int32)
        cmap[idx.internal_model.input.prognostic] = idx.internal_model.output.prognostic.to(torch.int32)
        cmap = cmap.to(device)
        x_state = x.clone()

    def forward(inp):
        if graphed is not None:
            return graphed(inp)
        with torch.no_grad():
            return model(inp, group) if group is not None else model(inp)

    def step():
        if cmap is None:
            return forward(x)
        x_state.copy_(x)  # every timed step starts from the same analysis
        if group is not None:  # BASELINE config 4 on N GPUs: the state stays sharded between the lead times (one small
            # grid-halo all-to-all-v per step), the forecast is all-gathered once, at the last lead time
            from anemoi_models_amd.distributed.partition import advance_sharded_state, sharded_forward
            from anemoi_models_amd.distributed.partition import sharded_state_output

            with torch.no_grad():
                for lead in range(args.rollout):
                    y_local, sp = sharded_forward(model, x_state, group, local_output=True)
                    if lead + 1 < args.rollout:
                        advance_sharded_state(model, x_state, y_local, sp, cmap)
                return sharded_state_output(model, x_state, y_local, sp, group)
        for lead in range(args.rollout):
            y = forward(x_state)
            if lead + 1 < args.rollout:
                ops.advance_input(x_state, y, cmap)
        return y

    # N > 1: before anything is timed, the partitioned forward is held against the SAME forward unsharded on one GPU
    # (rank 0 runs it; every rank's partitioned call returns the whole gathered output, so every rank checks its own copy
    # against rank 0's reference: a wrong halo list, gather offset or edge shard shows here, not as a fast wrong number)
    parity_vs_single = None
    if group is not None:
        import torch.distributed as dist

        stage[0] = "parity_vs_single"
        with torch.no_grad():
            y_part = model(x, group).float()
            y_one = model(x).float() if rank == 0 else torch.empty_like(y_part)
        if _backend_name(group) == "gloo":
            y_host = y_one.cpu()
            dist.broadcast(y_host, src=0)
            y_one = y_host.to(device)
        else:
            dist.broadcast(y_one, src=0)
        err = (y_part - y_one).abs().amax() / y_one.abs().amax().clamp_min(1e-30)
        finite = torch.isfinite(y_part).all().float()
        both = torch.stack([err.double(), 1.0 - finite.double()])
        if _backend_name(group) == "gloo":
            both_h = both.cpu()
            dist.all_reduce(both_h, op=dist.ReduceOp.MAX)
            both = both_h
        else:
            dist.all_reduce(both, op=dist.ReduceOp.MAX)
        parity_vs_single = {
            "max_rel_err": float(both[0]), "rows_checked": int(y_part.shape[-2]), "columns": int(y_part.shape[-1]),
            "ranks_checked": world, "finite": bool(float(both[1]) == 0.0),
            "vs": "the same forward unsharded on rank 0's GPU, max |a - b| / max |b| over the whole gathered output, "
                  "worst rank", "bound": PARITY_BOUND[args.dtype],
        }
        del y_part, y_one
        if not (parity_vs_single["finite"] and parity_vs_single["max_rel_err"] <= PARITY_BOUND[args.dtype]):
            # every rank holds the all-reduced figure: nothing is timed on a partition that computes something else
            dist.destroy_process_group()
            if rank == 0:
                print(json.dumps({"error": f"parity: partitioned forward differs from the single-GPU forward by "
                                           f"{parity_vs_single['max_rel_err']:.3e} (bound {PARITY_BOUND[args.dtype]:g})",
                                  "parity_vs_single": parity_vs_single, "stage": "parity_vs_single", "rank": 0}), flush=True)
            return 3
    stage[0] = "warmup"

    for _ in range(args.warmup):
        step()
    stage[0] = "timed steps"

    def fence():
        torch.cuda.synchronize()
        if group is not None:
            import torch.distributed as dist

            dist.barrier()
            torch.cuda.synchronize()

    # per-step marks on the launch stream (an event record costs < 1 us of host time and nothing on the device): the
    # median step time of SURVEY section 8d next to the mean the K-step bracket gives
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    fence()
    t0 = time.perf_counter()
    marks[0].record()
    y_timed = None
    for i in range(args.steps):
        y_timed = step()
        marks[i + 1].record()
    fence()
    elapsed = time.perf_counter() - t0
    # reproducibility of the step (no kernel of the path uses atomics: identical calls give identical bits): the last timed
    # output against two more steps, outside the timed region.  Reported, not gated -- a `false` is a hazard or a race.
    run_to_run_identical = None
    try:
        if isinstance(y_timed, torch.Tensor):
            kept = y_timed.clone()
            run_to_run_identical = bool(all(torch.equal(kept, step()) for _ in range(2)))
            del kept
        fence()
    except Exception:  # noqa: BLE001 -- the check must never cost the measurement
        run_to_run_identical = None
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    median_ms = per_step[len(per_step) // 2] if len(per_step) % 2 else 0.5 * (per_step[len(per_step) // 2 - 1]
                                                                             + per_step[len(per_step) // 2])
    if group is not None:
        import torch.distributed as dist

        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        if _backend_name(group) == "gloo":
            t = t.cpu()
        every = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        rank_ms = [float(e.item()) / args.steps * 1e3 for e in every]
        elapsed = max(float(e.item()) for e in every)  # the contract's MAX over ranks
    stage[0] = "exchange timing"
    exchanges = exchange_timing(model, x, group, elapsed / args.steps * 1e3, share_gpu) if group is not None else None
    stage[0] = "profile pass"

    ms_per_step = elapsed / args.steps * 1e3
    value = n_mesh * layers * args.rollout / (elapsed / args.steps)
    extra = profile_pass(model, x, group, args.dtype, args.detail and rank == 0,
                         traffic_ok=args.workload == "cfg3" and args.processor == "GraphTransformer", step_ms=ms_per_step)

    if rank == 0:
        line = {
            "metric": "mesh-node updates/sec (fwd step)", "value": round(value, 1), "unit": "mesh-node updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "ms_per_step_median": round(median_ms, 3),
            "value_at_median": round(n_mesh * layers * args.rollout / (median_ms * 1e-3), 1),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "bf16" if args.dtype == "bf16" else "f32", "data": "synthetic",
            "run_to_run_identical": run_to_run_identical,
            "config": {
                "workload": WORKLOADS[args.workload][4] + ", batch 1, 2 x 90 input vars -> 80 output vars, "
                                                          "full encoder+processor+decoder forward",
                "mesh_nodes": n_mesh, "grid_nodes": graph["data"].num_nodes, "processor_blocks": layers,
                "rollout_steps": args.rollout,
                "parallelism": "single GPU" if world == 1 else f"mesh node-partitioned over {world} GPUs, halo all-to-all-v"
                               + (" [DEBUG: ranks share one GPU, host-staged gloo; not a measurement]" if share_gpu else ""),
                "device": torch.cuda.get_device_name(local_rank),
            },
        }
        line.update(extra)
        if group is not None:
            line["rccl_ranks"] = rccl_ranks
            line["collective_backend"] = "gloo (host-staged, debug)" if share_gpu else "nccl (RCCL)"
            line["ms_per_step_ranks"] = {"min": round(min(rank_ms), 3), "max": round(max(rank_ms), 3),
                                         "all": [round(v, 3) for v in rank_ms]}
            line["parity_vs_single"] = parity_vs_single
            if exchanges is not None:
                line["exchanges"] = exchanges
        if group is not None:  # what this rank puts on the wire per forward step (rank 0's numbers; xGMI all-to-all-v)
            sp = next((v for k, v in model._idx_cache.items() if isinstance(k, tuple) and k[0] == "shard_plan"), None)
            if sp is not None:
                esz = 2 if args.dtype == "bf16" else 4
                c = model.num_channels
                proc_rows = sum(sp.proc.halo.send_splits) if sp.proc.halo is not None else 0
                dec_rows = sum(sp.dec.halo.send_splits) if sp.dec.halo is not None else 0
                gather_bytes = max(sp.dec_counts) * model.num_output_channels * 4
                line["halo"] = {
                    "exchanges_per_step": layers + (1 if sp.dec.halo is not None else 0), "own_mesh_rows": sp.hi - sp.lo,
                    "processor_rows_sent_per_exchange": proc_rows, "processor_rows_received_per_exchange": sp.proc.halo.n_recv
                    if sp.proc.halo is not None else 0, "decoder_rows_sent": dec_rows,
                    "bytes_sent_per_exchange": proc_rows * 2 * c * esz,
                    "bytes_sent_per_step": (proc_rows * layers + dec_rows) * 2 * c * esz * args.rollout,
                    "output_all_gather_bytes_per_rank": gather_bytes, "note": "k|v rows of boundary mesh nodes (2C values "
                    "per row), one all_to_all_single per processor block + one for the decoder; rank 0 of the group",
                }
        if args.processor != "GraphTransformer":
            line["config"]["workload"] = line["config"]["workload"].replace("GT blocks", f"{args.processor} blocks")
        if not args.no_cpu_baseline and world == 1 and args.processor == "GraphTransformer":
            stage[0] = "cpu baseline"
            n_cpu = args.cpu_blocks if args.cpu_blocks is not None else layers
            # the device's encoder output (mesh latent, internal Morton row order -> external node order) and prediction
            # for the same input, for the parity figures next to the baseline; on a bf16 run the exact-f32 route of the same
            # model as well (north_star: "within 1e-3 rel fp32" -- the f32 leg is what that sentence is checked on)
            hip_y, hip_latent = device_forward_with_latent(model, x)
            f32 = f32_leg(model, x) if args.dtype == "bf16" else None
            line["cpu_baseline"] = cpu_baseline(model, graph, x, idx, n_cpu, hip_latent, hip_y, f32)
        # every bound is evaluated BEFORE anything is printed: a run whose output failed its comparison prints the error
        # line only (with the measured errors), never a result line a reader of the first "{" would take for a number
        bound = PARITY_BOUND[args.dtype]
        base = line.get("cpu_baseline", {})
        par, par32 = base.get("parity", {}), base.get("parity_fp32", {})
        checks = [("output_rel_err", par.get("output_rel_err"), bound),
                  ("encoder_out_rel_err", par.get("encoder_out_rel_err"), LATENT_BOUND[args.dtype]),
                  ("parity_fp32.output_rel_err", par32.get("output_rel_err"), PARITY_BOUND["fp32"]),
                  ("parity_fp32.per_variable_rel_err_max", par32.get("per_variable_rel_err_max"), PARITY_BOUND["fp32"]),
                  ("parity_fp32.encoder_out_rel_err", par32.get("encoder_out_rel_err"), LATENT_BOUND["fp32"])]
        bad = [f"{name} {val:.3e} > {lim:g}" for name, val, lim in checks if val is not None and not val <= lim]
        if bad:
            failure = {"error": "parity against the CPU oracle: " + "; ".join(bad), "parity": par, "parity_fp32":
                       {k: v for k, v in par32.items() if k != "per_variable_rel_err"}}
        if parity_vs_single is not None and not (parity_vs_single["finite"] and parity_vs_single["max_rel_err"] <= bound):
            failure = {"error": f"parity: partitioned forward differs from the single-GPU forward by "
                                f"{parity_vs_single['max_rel_err']:.3e} (bound {bound:g})", "parity_vs_single": parity_vs_single}
        if (failure is None and world == 1 and not args.no_secondary and args.workload == "cfg3" and args.dtype == "bf16"
                and args.processor == "GraphTransformer" and args.rollout == 1 and not args.hipgraph):
            stage[0] = "secondary"
            graph_cfg3 = graph
            del model, x, graphed, y_timed
            torch.cuda.empty_cache()
            line["secondary"] = secondary_block(device, args.dtype, graph_cfg3)
        if failure is None:
            print(json.dumps(line), flush=True)
    stage[0] = "teardown"
    if group is not None:
        import torch.distributed as dist

        dist.destroy_process_group()
    if failure is not None:
        print(json.dumps({**failure, "stage": "parity", "rank": rank}), flush=True)
        return 3
    return 0


def exchange_timing(model, x, group, step_ms: float, share_gpu: bool):
    """N > 1: what the halo exchanges cost and how much of it the step sees.

    * ``alone_us``: every all-to-all-v of the step's plans (processor k|v halo, decoder halo) issued ALONE, device idle on
      both sides, 20 repetitions -- the collective's own latency at this world size (min / median / max over the
      repetitions, worst rank);
    * ``in_step``: one instrumented forward with device events on the launch stream around ``HaloExchange.start`` (pack
      + enqueue) and ``HaloExchange.finish`` (the stream waiting for the transfer): ``exposed_us`` is the time the
      compute stream stood still in ``finish`` -- 0 when the transfer hides behind the GEMM launched in between --,
      ``window_us`` start to finish.  ``exposed_fraction_of_step`` = sum of the exposed waits / the measured step.
    """
    import statistics

    import torch.distributed as dist

    from anemoi_models_amd.distributed.partition import HaloExchange

    sp = next((v for k, v in model._idx_cache.items() if isinstance(k, tuple) and k[0] == "shard_plan"), None)
    if sp is None:
        return None
    dtype = torch.bfloat16 if os.environ.get("ANEMOI_AMD_DTYPE") == "bf16" else torch.float32
    width = 2 * model.num_channels
    out = {}

    def reduce_max(vals):
        t = torch.tensor(vals, dtype=torch.float64, device="cpu" if share_gpu else x.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        return [float(v) for v in t.tolist()]

    alone = {}
    for name, lg in (("processor", sp.proc), ("decoder", sp.dec)):
        halo = getattr(lg, "halo", None)
        if halo is None:
            continue
        n_own = lg.n_own_src
        rows = torch.zeros((n_own + halo.n_recv, width), dtype=dtype, device=x.device)
        us = []
        for it in range(23):
            torch.cuda.synchronize()
            dist.barrier(group=group)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            halo.exchange(rows, n_own)
            torch.cuda.synchronize()
            if it >= 3:
                us.append((time.perf_counter() - t0) * 1e6)
        lo, med, hi = reduce_max([min(us), statistics.median(us), max(us)])
        alone[name] = {"min": round(lo, 1), "median": round(med, 1), "max": round(hi, 1),
                       "rows_sent": int(sum(halo.send_splits)), "rows_received": int(halo.n_recv),
                       "bytes_received": int(halo.n_recv) * width * rows.element_size()}
        del rows
    out["alone_us"] = alone
    HaloExchange.TIMING = []
    try:
        with torch.no_grad():
            model(x, group)
        torch.cuda.synchronize()
        rec = HaloExchange.TIMING
    finally:
        HaloExchange.TIMING = None
    if rec:
        start = [e0.elapsed_time(e1) * 1e3 for e0, e1, e2, e3 in rec]
        exposed = [e2.elapsed_time(e3) * 1e3 for e0, e1, e2, e3 in rec]
        window = [e0.elapsed_time(e3) * 1e3 for e0, e1, e2, e3 in rec]

        def mmm(v):
            return {"min": round(min(v), 1), "median": round(statistics.median(v), 1), "max": round(max(v), 1)}

        tot = reduce_max([sum(exposed), sum(start)])
        out["in_step"] = {"exchanges": len(rec), "start_us": mmm(start), "exposed_us": mmm(exposed), "window_us": mmm(window),
                          "exposed_us_per_step_worst_rank": round(tot[0], 1), "pack_us_per_step_worst_rank": round(tot[1], 1),
                          "exposed_fraction_of_step": round(tot[0] * 1e-3 / step_ms, 4), "rank": 0}
    out["note"] = ("device events on the launch stream around HaloExchange.start / finish; 'exposed' = the compute stream "
                   "waiting inside finish" + (" [DEBUG: host-staged gloo on a shared GPU -- not RCCL timings]" if share_gpu else ""))
    return out


def _backend_name(group) -> str:
    import torch.distributed as dist

    return dist.get_backend(group)


def main() -> int:
    """Runs the benchmark; any failure (process-group init, a collective, a kernel status, parity beyond its bound) ends in
    ONE JSON error line on stdout and a non-zero exit code -- never a bare traceback the driver would have to parse, never
    a number printed for an output nobody compared with anything."""
    stage = Stage()
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus, sys.argv[1:])  # nothing has touched the GPU yet
    try:
        return _run(stage, args)
    except SystemExit:
        raise
    except BaseException as exc:  # noqa: BLE001  (KeyboardInterrupt included: the driver's timeout)
        import traceback

        traceback.print_exc(file=sys.stderr)
        print(json.dumps({"error": f"{type(exc).__name__}: {exc}"[:600], "stage": stage[0],
                          "rank": int(os.environ.get("RANK", "0"))}), flush=True)
        return 2


if __name__ == "__main__":
    sys.exit(main())
