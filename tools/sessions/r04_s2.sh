#!/bin/bash
# Round 4, GPU session 2: attention at mesh size vs f64 reference, bench.py N > 1 contract (ranks sharing the GPU)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_s2
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python3 -m pytest tests/test_gpu_attention_sizes.py -x -q -m gpu -s > "$OUT/pytest_attention_sizes.txt" 2>&1
tail -8 "$OUT/pytest_attention_sizes.txt"
timeout 1200 python3 -m pytest tests/test_bench_contract.py -x -q -m gpu > "$OUT/pytest_bench_contract.txt" 2>&1
tail -8 "$OUT/pytest_bench_contract.txt"
