#!/bin/bash
# round 5, session 50: attention stress (ragged shapes, fresh inputs, shifting allocations, against torch f32) and the op fuzzers on the final library, with a second process on the GPU
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s50
mkdir -p "$OUT"
cd "$ROOT"
timeout 300 python3 tools/micro/ops_repeat.py linear 2000000 > "$OUT/noise.txt" 2>&1 &
NOISE=$!
timeout 150 python3 tools/mhsa_stress.py 80 > "$OUT/stress.txt" 2>&1; grep -v amdgpu.ids "$OUT/stress.txt" | tail -n 6 | cut -c1-200
timeout 120 python3 tools/fuzz_ops.py 60 7 > "$OUT/fuzz.txt" 2>&1; grep -v amdgpu.ids "$OUT/fuzz.txt" | tail -n 8 | cut -c1-200
kill $NOISE 2>/dev/null; wait $NOISE 2>/dev/null
