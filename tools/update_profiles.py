#!/usr/bin/env python
"""Copy the output of tools/refresh_profiles.sh (gpurun_out/profiles_<tag>/) into the tracked profiles/ directory.

Usage: python tools/update_profiles.py r01
Builds profiles/<tag>_traffic.json from the two PMC passes (FETCH_SIZE counts 32-byte... see the "_source" note:
on gfx950 FETCH_SIZE reports half of the bytes for 16 B/lane coalesced reads, WRITE_SIZE is exact; both in KB).
"""
import json
import os
import re
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"profiles_{tag}")
dst = os.path.join(root, "profiles")

for name, out in [("bench_cfg1_bf16.json", f"{tag}_bench_cfg1_bf16.json"),
                  ("bench_cfg2_bf16.json", f"{tag}_bench_cfg2_bf16.json"),
                  ("bench_cfg3_bf16.json", f"{tag}_bench_cfg3_bf16.json"),
                  ("bench_cfg3_fp32.json", f"{tag}_bench_cfg3_fp32.json"),
                  ("bench_cfg4_rollout4_bf16.json", f"{tag}_bench_cfg4_rollout4_bf16.json"),
                  ("bench_cfg5_gnn_bf16.json", f"{tag}_bench_cfg5_gnn_bf16.json"),
                  ("bench_cfg3_transformer_bf16.json", f"{tag}_bench_cfg3_transformer_bf16.json"),
                  ("bench_cfg2_transformer_bf16.json", f"{tag}_bench_cfg2_transformer_bf16.json"),
                  ("bench_cfg2_bf16_cpu_baseline.json", f"{tag}_bench_cfg2_bf16_cpu_baseline.json"),
                  ("kernel_summary_transformer.txt", f"{tag}_bench_cfg3_transformer_summary.txt"),
                  ("pmc_mhsa_mfma_busy.txt", f"{tag}_mhsa_mfma_busy_pmc.txt"),
                  ("kernel_stats.csv", f"{tag}_bench_cfg3_bf16_kernel_stats.csv"),
                  ("kernel_summary.txt", f"{tag}_bench_cfg3_bf16_summary.txt"),
                  ("bench_cfg3_bf16_detail.txt", f"{tag}_bench_cfg3_bf16_per_shape.txt")]:
    p = os.path.join(src, name)
    if os.path.exists(p):
        shutil.copyfile(p, os.path.join(dst, out))


def parse(path):
    rows = {}
    pat = re.compile(r"^(.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+n=\s*(\d+)\s+mean=\s*([0-9.]+)")
    for line in open(path):
        m = pat.match(line.rstrip())
        if m:
            rows[m.group(1).strip()] = (int(m.group(3)), float(m.group(4)))
    return rows


fetch, write = parse(os.path.join(src, "pmc_fetch_size.txt")), parse(os.path.join(src, "pmc_write_size.txt"))


def merged(prefix):
    n = f = w = 0.0
    names = []
    for k, (cnt, val) in fetch.items():
        if prefix in k and k in write:
            n += cnt
            f += cnt * val
            w += cnt * write[k][1]
            names.append(k)
    return {"kernel": prefix, "instantiations": names, "launches": int(n), "FETCH_SIZE_kb": round(f / n, 1),
            "WRITE_SIZE_kb": round(w / n, 1), "traffic_bytes_per_launch": int((2 * f / n + w / n) * 1024)}


def one(prefix):
    for k, (cnt, val) in fetch.items():
        if prefix in k:
            return {"FETCH_SIZE": val, "WRITE_SIZE": write[k][1]}
    return None


out = {
    "_source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on `python3 bench.py "
               "--steps 2 --warmup 1 --no-cpu-baseline` (config 3, bf16), mean per dispatch over all launches of the "
               "kernel (tools/refresh_profiles.sh). Corrections per MI355X_MICROARCH.md section HBM, calibrated on this "
               "repo's own add_kernel / layer_norm_kernel (known byte counts, 16 B/lane coalesced): FETCH_SIZE reads "
               "exactly 1/2 of the bytes (x2), WRITE_SIZE is exact (x1); both in KB.",
    "calibration": {"add_kernel": dict(known_read_kb=163848, known_write_kb=81924, **(one("add_kernel") or {})),
                    "layer_norm_kernel": dict(known_read_kb=184800, known_write_kb=184800,
                                              **(one("layer_norm_kernel") or {}))},
    "kernels": {"linear": merged("linear_bf16_w4_kernel"), "gt_edge_attention": merged("gt_edge_attention_folded_")},
}
with open(os.path.join(dst, f"{tag}_traffic.json"), "w") as fh:
    json.dump(out, fh, indent=1)
print(json.dumps(out["kernels"], indent=1))
