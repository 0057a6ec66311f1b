#!/bin/bash
# round 6, session 22: staggered GEMM start -- phases / unit scan (informative on a box whose "off" is >= 35.5 ms)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_s23_$$
mkdir -p "$OUT"
cd "$ROOT"
for i in 1 2; do
  for cfg in 0,0,2 2,16,2 2,8,2 2,12,2 2,24,2 2,32,2 2,48,2 2,0,2,12 2,0,2,25 2,0,2,40 2,0,2,50 3,12,2 4,8,2; do
    ANEMOI_AMD_GEMM_STAGGER=$cfg timeout 300 python3 bench.py --no-cpu-baseline --no-secondary > "$OUT/bench_${cfg}_$i.json" 2>/dev/null
    echo "stagger $cfg run $i: $(grep -o '"ms_per_step": [0-9.]*' "$OUT/bench_${cfg}_$i.json" | head -1) $(grep -o '"linear": [0-9.]*' "$OUT/bench_${cfg}_$i.json" | head -1)"
  done
done
