#!/bin/bash
# round 5, session 47: attention backward timing after its wait statements took their accumulators as operands
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s47
mkdir -p "$OUT"
cd "$ROOT"
timeout 400 python3 tools/mhsa_bwd_bench.py > "$OUT/bwd.txt" 2>&1; grep -v amdgpu.ids "$OUT/bwd.txt" | tail -n 8 | cut -c1-220
