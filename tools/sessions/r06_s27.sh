#!/bin/bash
# round 6, session 27: the staggered GEMM start on a warm and on a cold weight panel (mechanism), fc1's shape
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_s27
mkdir -p "$OUT"
cd "$ROOT"
for i in 1 2; do for cfg in 0,0,2 2,16,2; do
  ANEMOI_AMD_GEMM_STAGGER=$cfg timeout 200 python3 tools/micro/gemm_cold_w.py 2>/dev/null | tee -a "$OUT/cold_w.txt"
done; done
ANEMOI_AMD_GEMM_STAGGER=0,0,2 timeout 200 python3 bench.py --no-cpu-baseline --no-secondary 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1
