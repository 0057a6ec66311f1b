#!/bin/bash
# round 6, session 21: staggered GEMM start (shipped default 4,8,2) against off, alternating on one box; the tile-kernel model test
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_s21_$$
mkdir -p "$OUT"
cd "$ROOT"
for i in 1 2 3 4; do
  for cfg in 0,0,2 4,8,2; do
    ANEMOI_AMD_GEMM_STAGGER=$cfg timeout 300 python3 bench.py --no-cpu-baseline --no-secondary > "$OUT/bench_${cfg}_$i.json" 2>/dev/null
    echo "stagger $cfg run $i: $(grep -o '"ms_per_step": [0-9.]*' "$OUT/bench_${cfg}_$i.json" | head -1) $(grep -o '"linear": [0-9.]*' "$OUT/bench_${cfg}_$i.json" | head -1)"
  done
done
timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "tile_edge_kernel or float16_autocast or tiles_is_the_plain" 2>&1 | tail -n 3
