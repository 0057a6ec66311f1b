#!/usr/bin/env python
"""Node-partitioned forward / training step at world sizes and shapes the fixed tests do not cover (5, 6, 7, 8 ranks; every
family), the ranks sharing this GPU (tests/_gpu_shared_ranks.py: HIP kernels on cuda:0, host-staged gloo): sharded output ==
unsharded output, summed sharded gradients == single-device gradients.  python tools/micro/partition_sweep.py"""
import os
import subprocess
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
worker = os.path.join(ROOT, "tests", "_gpu_shared_ranks.py")
CASES = [
    (5, "o32_ico2", 64, 2, 16, "fp32", "GraphTransformer", False), (7, "o48_ico3", 128, 2, 8, "bf16", "GraphTransformer", False),
    (8, "o48_ico3", 256, 2, 16, "bf16", "GraphTransformer", False), (6, "o32_ico2", 64, 2, 4, "bf16", "GNN_all", False),
    (5, "o48_ico3", 128, 2, 16, "fp32", "GNN_all", False), (4, "o32_ico2", 128, 2, 4, "bf16", "Transformer", False),
    (8, "o32_ico2", 128, 2, 8, "fp32", "Transformer", False), (3, "o32_ico2", 64, 2, 16, "fp32", "GraphTransformer", True),
    (5, "o48_ico3", 128, 2, 8, "bf16", "GraphTransformer", True), (4, "o32_ico2", 64, 2, 16, "fp32", "GNN_all", True),
    (3, "o32_ico2", 128, 2, 4, "fp32", "Transformer", True), (6, "o48_ico3", 192, 2, 16, "bf16", "GraphTransformer", False),
]
if len(sys.argv) > 1:  # python tools/micro/partition_sweep.py 3,o32_ico2,128,2,4,fp32,Transformer,train ...
    CASES = []
    for a in sys.argv[1:]:
        w, gname, c, l_, h, dt, fam, tr = a.split(",")
        CASES.append((int(w), gname, int(c), int(l_), int(h), dt, fam, tr == "train"))
bad = 0
for n, (world, graph, channels, layers, heads, dtype, family, train) in enumerate(CASES):
    port = 29900 + (os.getpid() + n) % 90
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "res")
        env = dict(os.environ, ANEMOI_TEST_FAMILY=family)
        args = [str(world), str(port), out, graph, str(channels), str(layers), str(heads), dtype] + (["train"] if train else [])
        procs = [subprocess.Popen([sys.executable, worker, str(r)] + args, env=env, stdout=subprocess.DEVNULL,
                                  stderr=subprocess.PIPE) for r in range(world)]
        codes, errs = [], []
        for p in procs:
            try:
                _, e = p.communicate(timeout=600)
                codes.append(p.returncode)
                errs.append(e.decode()[-400:])
            except subprocess.TimeoutExpired:
                p.kill()
                codes.append(-9)
                errs.append("timeout")
        what = f"world {world} {graph} C={channels} L={layers} H={heads} {dtype} {family} {'train' if train else 'forward'}"
        if codes != [0] * world:
            bad += 1
            print(f"{what}: exit codes {codes}\n    {[e.splitlines()[-1] if e.splitlines() else '' for e in errs][:2]}", flush=True)
            continue
        infos = [torch.load(f"{out}.{r}") for r in range(world)]
        tol = 2e-5 if dtype == "fp32" else 3e-2
        worst = max(i["err"] / max(1.0, i["scale"]) for i in infos)
        g_worst = max((i.get("grad_err", 0.0) / max(i.get("grad_scale", 1.0), 1e-30) for i in infos), default=0.0)
        ok = all(i["finite"] and i["rerun"] == 0.0 for i in infos) and worst <= tol
        bad += not ok
        print(f"{what}: {'ok' if ok else 'BAD'}  sharded vs unsharded {worst:.2e}" + (f", gradients {g_worst:.2e}" if train else "")
              + f" (rows per rank {[i['own'] for i in infos]})", flush=True)
print(f"{bad} bad of {len(CASES)}")
