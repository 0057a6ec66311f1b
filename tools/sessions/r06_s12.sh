#!/bin/bash
# round 6, session 12: staggered GEMM start, runtime switch, on/off alternating on ONE box (the gain differed between the boxes of s10 and s11)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_s12_$(hostname | tr -c 'a-zA-Z0-9' '_')$$
mkdir -p "$OUT"
cd "$ROOT"
for i in 1 2 3; do
  for cfg in 0,0,3 4,8,3 4,8,2 4,8,1; do
    ANEMOI_AMD_GEMM_STAGGER=$cfg timeout 300 python3 bench.py --no-cpu-baseline --no-secondary > "$OUT/bench_${cfg}_$i.json" 2>/dev/null
    echo "stagger $cfg run $i: $(grep -o '"ms_per_step": [0-9.]*' "$OUT/bench_${cfg}_$i.json" | head -1) $(grep -o '"linear": [0-9.]*' "$OUT/bench_${cfg}_$i.json" | head -1)"
  done
  timeout 300 python3 tools/micro/bench_with_lib.py tools/micro/bin/libanemoi_amd_stag4x1.so --no-cpu-baseline --no-secondary > "$OUT/bench_lab4x1_$i.json" 2>/dev/null
  echo "lab stag4x1 run $i: $(grep -o '"ms_per_step": [0-9.]*' "$OUT/bench_lab4x1_$i.json" | head -1) $(grep -o '"linear": [0-9.]*' "$OUT/bench_lab4x1_$i.json" | head -1)"
done
