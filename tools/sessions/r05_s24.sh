#!/bin/bash
# round 5, session 24: K = 256 row-streaming GEMM kernel -- test, micro timing against the four-wave kernel, bench
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s24
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "k256 or linear" > "$OUT/pytest_k256.txt" 2>&1; grep -a "passed\|failed\|Error\|assert" "$OUT/pytest_k256.txt" | tail -8
cat > /tmp/k256_bench.py <<'P'
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from anemoi_models_amd import ops
dev = "cuda"
m, k = 542080, 256
x = torch.randn(m, k, device=dev).bfloat16()
for n, fold in ((2048, True), (2240, True), (1024, False), (256, False)):
    w = (torch.randn(n, k, device=dev) / 16).bfloat16()
    b = torch.randn(n, device=dev)
    ln = (torch.rand(m, 2, device=dev).contiguous(), torch.randn(n, device=dev)) if fold else None
    y = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    for _ in range(3): ops.linear(x, w, b, ln=ln, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.linear(x, w, b, ln=ln, out=y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"K=256 M={m} N={n} fold={fold}: {ms:.4f} ms  {2*m*n*k/ms/1e9:.0f} TFLOP/s  output {m*n*2/ms/1e6:.0f} GB/s", flush=True)
P
{
echo "== row-streaming kernel"; python3 /tmp/k256_bench.py
echo "== four-wave kernel (ANEMOI_AMD_GEMM_KSTREAM=0)"; ANEMOI_AMD_GEMM_KSTREAM=0 python3 /tmp/k256_bench.py
} > "$OUT/k256_ab.txt" 2>&1; grep -v amdgpu "$OUT/k256_ab.txt"
python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; cut -c1-260 "$OUT/bench_default.json"; tail -2 "$OUT/bench_default.err"
ANEMOI_AMD_GEMM_KSTREAM=0 python3 bench.py > "$OUT/bench_off.json" 2> "$OUT/bench_off.err"; cut -c1-260 "$OUT/bench_off.json"
