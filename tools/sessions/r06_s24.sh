#!/bin/bash
# round 6, session 24: staggered GEMM start -- who forms a phase (position in the XCD / the XCD / both), small-problem launches (config 2)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_s24_$$
mkdir -p "$OUT"
cd "$ROOT"
for i in 1 2; do
  for cfg in 0,0,2 2,16,2,0,0 2,16,2,0,1 2,16,2,0,2 4,8,2,0,2 8,4,2,0,1 2,16,1,0,0; do
    ANEMOI_AMD_GEMM_STAGGER=$cfg timeout 300 python3 bench.py --no-cpu-baseline --no-secondary > "$OUT/bench_${cfg}_$i.json" 2>/dev/null
    echo "stagger $cfg run $i: $(grep -o '"ms_per_step": [0-9.]*' "$OUT/bench_${cfg}_$i.json" | head -1) $(grep -o '"linear": [0-9.]*' "$OUT/bench_${cfg}_$i.json" | head -1)"
  done
  for cfg in 0,0,2 2,16,2,0,0,1 2,16,2,0,0,2 2,16,2,0,0,4 4,8,2,0,0,1; do
    ANEMOI_AMD_GEMM_STAGGER=$cfg timeout 300 python3 bench.py --no-cpu-baseline --no-secondary --workload cfg2 --steps 50 --warmup 10 > "$OUT/bench_cfg2_${cfg}_$i.json" 2>/dev/null
    echo "cfg2 stagger $cfg run $i: $(grep -o '"ms_per_step": [0-9.]*' "$OUT/bench_cfg2_${cfg}_$i.json" | head -1) $(grep -o '"linear": [0-9.]*' "$OUT/bench_cfg2_${cfg}_$i.json" | head -1)"
  done
done
