#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s8
mkdir -p "$OUT"
cd "$ROOT"
timeout 1700 python3 -m pytest tests -q -m gpu > "$OUT/pytest_gpu.txt" 2>&1; tail -6 "$OUT/pytest_gpu.txt"
