#!/bin/bash
# round 6, session 11: staggered GEMM start -- scan of (phases, unit, min rounds) on the whole model, alternating with "off"
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_s11
mkdir -p "$OUT"
cd "$ROOT"
for i in 1 2; do
  for cfg in 0,0,3 4,8,3 4,4,3 8,4,3 2,8,3 4,12,3 8,2,3 4,8,6 4,8,2 16,2,3; do
    ANEMOI_AMD_GEMM_STAGGER=$cfg timeout 300 python3 bench.py --no-cpu-baseline --no-secondary > "$OUT/bench_${cfg}_$i.json" 2>/dev/null
    echo "stagger $cfg run $i: $(grep -o '"ms_per_step": [0-9.]*' "$OUT/bench_${cfg}_$i.json" | head -1) $(grep -o '"linear": [0-9.]*' "$OUT/bench_${cfg}_$i.json" | head -1) identical $(grep -o '"run_to_run_identical": [a-z]*' "$OUT/bench_${cfg}_$i.json")"
  done
done
for cfg in 0,0,3 4,8,3; do
  ANEMOI_AMD_GEMM_STAGGER=$cfg timeout 300 python3 bench.py --no-cpu-baseline --no-secondary --workload cfg2 > "$OUT/bench_cfg2_${cfg}.json" 2>/dev/null; echo "cfg2 stagger $cfg: $(grep -o '"ms_per_step": [0-9.]*' "$OUT/bench_cfg2_${cfg}.json" | head -1)"
  ANEMOI_AMD_GEMM_STAGGER=$cfg timeout 300 python3 bench.py --no-cpu-baseline --no-secondary --processor Transformer --steps 5 --warmup 2 > "$OUT/bench_tfm_${cfg}.json" 2>/dev/null; echo "tfm cfg3 stagger $cfg: $(grep -o '"ms_per_step": [0-9.]*' "$OUT/bench_tfm_${cfg}.json" | head -1)"
done
