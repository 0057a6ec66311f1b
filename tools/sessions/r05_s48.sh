#!/bin/bash
# round 5, session 48: run-to-run determinism of the three model families (forward and a training step) with a second process computing on the GPU
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s48
mkdir -p "$OUT"
cd "$ROOT"
timeout 400 python3 tools/micro/ops_repeat.py linear 2000000 > "$OUT/noise.txt" 2>&1 &
NOISE=$!
timeout 330 python3 tools/determinism_check.py cfg2 60 > "$OUT/cfg2.txt" 2>&1
grep -v amdgpu.ids "$OUT/cfg2.txt" | tail -n 8 | cut -c1-200
timeout 200 python3 tools/determinism_check.py cfg3 12 > "$OUT/cfg3.txt" 2>&1
grep -v amdgpu.ids "$OUT/cfg3.txt" | tail -n 8 | cut -c1-200
if kill -0 $NOISE 2>/dev/null; then echo "the contending process ran throughout"; else echo "the contending process ENDED early"; fi
kill $NOISE 2>/dev/null; wait $NOISE 2>/dev/null
