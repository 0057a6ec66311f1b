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
  cpu_baseline  the CPU oracle (plain PyTorch, same algorithm as the reference) timed on this host's cores on a
                bounded sample of the same workload.
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
    ap.add_argument("--cpu-blocks", type=int, default=2, help="processor blocks in the CPU-baseline sample")
    ap.add_argument("--rollout", type=int, default=1,
                    help="autoregressive forecasts per step (BASELINE config 4 = --rollout 4): forward, then the "
                         "in-place input update anemoi_advance_input, repeated")
    ap.add_argument("--detail", action="store_true", help="print a per-shape kernel table to stderr")
    ap.add_argument("--hipgraph", action="store_true",
                    help="replay the forward as one captured HIP graph (single GPU; pays off on the small workloads)")
    return ap.parse_args()


def build(workload: str, device, processor: str = "GraphTransformer"):
    from anemoi_models_amd.graphs.synthetic import build_graph
    from anemoi_models_amd.models import AnemoiModelEncProcDec
    from anemoi_models_amd.utils.indices import SimpleDataIndices
    from anemoi_models_amd.utils.presets import model_config

    graph_name, channels, layers, heads, _ = WORKLOADS[workload]
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
        key = (name,) + tuple(sorted((k, v) for k, v in work.items() if k not in ("flops", "bytes")))
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


def profile_pass(model, x, group, dtype_name: str, detail: bool = False, traffic_ok: bool = True):
    """One instrumented forward: HIP events around every kernel launch, on the launch stream."""
    from anemoi_models_amd import ops

    ops.PROFILE = []
    with torch.no_grad():
        model(x, group) if group is not None else model(x)
    torch.cuda.synchronize()
    records, ops.PROFILE = ops.PROFILE, None
    if detail:
        detail_table(records, dtype_name)
    agg = {}
    for name, start, end, work in records:
        a = agg.setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
        a["launches"] += 1
        a["ms"] += start.elapsed_time(end)
        a["flops"] += work.get("flops", 0)
        a["bytes"] += work.get("bytes", 0)
    out = {}
    if "linear" in agg and agg["linear"]["ms"] > 0:
        a = agg["linear"]
        achieved = a["flops"] / (a["ms"] * 1e-3) / 1e12
        peak = MFMA_PEAK_TFLOPS[dtype_name]
        out["roofline"] = {
            "kernel": "anemoi::linear_bf16_w4_kernel / linear_kernel (anemoi_linear: fused Linear, MFMA)", "bound": "mfma", "achieved": round(achieved, 2),
            "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": None,
            "launches": a["launches"], "avg_launch_ms": round(a["ms"] / a["launches"], 4),
            "flops_per_launch": a["flops"] / a["launches"], "bytes_per_launch": a["bytes"] / a["launches"],
            "share_of_step": None,
        }
    if "gt_edge_attention" in agg and agg["gt_edge_attention"]["ms"] > 0:
        a = agg["gt_edge_attention"]
        achieved = a["bytes"] / (a["ms"] * 1e-3) / 1e9
        out["roofline_edge"] = {
            "kernel": "anemoi::gt_edge_attention_kernel (fused gather/lin_edge/softmax/scatter)", "bound": "hbm",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "launches": a["launches"],
            "avg_launch_ms": round(a["ms"] / a["launches"], 4), "bytes_per_launch": a["bytes"] / a["launches"],
        }
    if "mhsa" in agg and agg["mhsa"]["ms"] > 0:  # Transformer processor: mesh-node self attention (MFMA-bound)
        a = agg["mhsa"]
        achieved = a["flops"] / (a["ms"] * 1e-3) / 1e12
        peak = MFMA_PEAK_TFLOPS[dtype_name]
        out["roofline_mhsa"] = {
            "kernel": "anemoi::mhsa_bf16_d64_kernel (anemoi_mhsa: flash attention on v_mfma_f32_32x32x16_bf16)",
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
    # HBM traffic per launch from the committed rocprofv3 PMC passes of this same command (PMC counters cannot be
    # collected from inside the process; see profiles/r01_traffic.json for the command and the gfx950 corrections)
    try:
        if not traffic_ok or group is not None:  # the PMC passes were taken on config 3 / bf16 / one GPU only
            raise KeyError("no PMC pass for this run")
        with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as f:
            traffic = json.load(f)["kernels"]
        if "roofline" in out and dtype_name == "bf16":
            out["roofline"]["traffic"] = traffic["linear"]["traffic_bytes_per_launch"]
            out["roofline"]["traffic_source"] = "profiles/r01_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE)"
        if "roofline_edge" in out and dtype_name == "bf16":
            out["roofline_edge"]["traffic"] = traffic["gt_edge_attention"]["traffic_bytes_per_launch"]
            out["roofline_edge"]["traffic_source"] = "profiles/r01_traffic.json"
    except (OSError, KeyError, ValueError):
        pass
    total_ms = sum(a["ms"] for a in agg.values())
    out["kernel_time_ms"] = {k: round(v["ms"], 3) for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}
    if "roofline" in out and total_ms > 0:
        out["roofline"]["share_of_step"] = round(agg["linear"]["ms"] / total_ms, 3)
    return out


def cpu_baseline(model, graph, n_blocks: int):
    """CPU oracle (plain PyTorch restatement of the reference algorithm) on a bounded sample of the workload.

    Sample: the first ``n_blocks`` GraphTransformer processor blocks at the full mesh size / channel width, f32, all
    host cores PyTorch uses.  The unit is the metric's: mesh-node updates per second over those blocks.
    """
    from oracle import reference_path as ref  # checker / baseline only

    p = model.processor
    sd = {"processor." + k: v.detach().float().cpu() for k, v in p.state_dict().items()}
    edge_attr = ref.trainable_tensor(p.edge_attr.cpu(), sd["processor.trainable.trainable"], 1)
    edge_index = p.edge_index_base.cpu()
    n_mesh, c = graph["hidden"].num_nodes, model.num_channels
    heads = p.proc[0].blocks[0].num_heads
    xm = torch.randn(n_mesh, c, generator=torch.Generator().manual_seed(3))
    blocks = [f"processor.proc.{ci}.blocks.{bi}" for ci in range(len(p.proc)) for bi in range(len(p.proc[ci].blocks))]
    blocks = blocks[:n_blocks]
    with torch.no_grad():
        ref.gt_processor_block(sd, blocks[0], xm[:256], edge_attr[:8], edge_index[:, :8] % 256, heads)  # warm-up
        t0 = time.perf_counter()
        for name in blocks:
            xm = ref.gt_processor_block(sd, name, xm, edge_attr, edge_index, heads)
        dt = time.perf_counter() - t0
    return {
        "value": round(n_mesh * len(blocks) / dt, 1), "unit": "mesh-node updates/s", "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": f"{len(blocks)} of {len(p.proc) * len(p.proc[0].blocks)} GraphTransformer processor blocks at full "
                  f"size ({n_mesh} mesh nodes, {c} ch, f32, CPU oracle = plain PyTorch), {dt:.1f} s; mappers excluded",
    }


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one process per GPU)")
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
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
        group = dist.group.WORLD

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
        for lead in range(args.rollout):
            y = forward(x_state)
            if lead + 1 < args.rollout:
                ops.advance_input(x_state, y, cmap)
        return y

    for _ in range(args.warmup):
        step()

    def fence():
        torch.cuda.synchronize()
        if group is not None:
            import torch.distributed as dist

            dist.barrier()
            torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if group is not None:
        import torch.distributed as dist

        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = elapsed / args.steps * 1e3
    value = n_mesh * layers * args.rollout / (elapsed / args.steps)
    extra = profile_pass(model, x, group, args.dtype, args.detail and rank == 0,
                         traffic_ok=args.workload == "cfg3" and args.processor == "GraphTransformer")

    if rank == 0:
        line = {
            "metric": "mesh-node updates/sec (fwd step)", "value": round(value, 1), "unit": "mesh-node updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "bf16" if args.dtype == "bf16" else "f32", "data": "synthetic",
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
        if args.processor != "GraphTransformer":
            line["config"]["workload"] = line["config"]["workload"].replace("GT blocks", f"{args.processor} blocks")
        if not args.no_cpu_baseline and world == 1 and args.processor == "GraphTransformer":
            line["cpu_baseline"] = cpu_baseline(model, graph, args.cpu_blocks)
        print(json.dumps(line), flush=True)
    if group is not None:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
