#!/bin/bash
# round 6, session 10: where does the staggered GEMM start gain inside the model?  per-shape tables, alternating libraries
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_s10
mkdir -p "$OUT"
cd "$ROOT"
B=tools/micro/bin
for i in 1 2 3; do
  for lib in ship stag2x1 stag4x1; do
    if [ $lib = ship ]; then timeout 300 python3 bench.py --no-cpu-baseline --no-secondary --detail > "$OUT/bench_${lib}_$i.json" 2> "$OUT/detail_${lib}_$i.txt"
    else timeout 300 python3 tools/micro/bench_with_lib.py $B/libanemoi_amd_$lib.so --no-cpu-baseline --no-secondary --detail > "$OUT/bench_${lib}_$i.json" 2> "$OUT/detail_${lib}_$i.txt"; fi
    echo "bench $lib $i: $(grep -o '"ms_per_step": [0-9.]*' "$OUT/bench_${lib}_$i.json" | head -1) $(grep -o '"ms_per_step_median": [0-9.]*' "$OUT/bench_${lib}_$i.json" | head -1) $(grep -o '"kernel_time_ms": {[^}]*}' "$OUT/bench_${lib}_$i.json")"
  done
done
