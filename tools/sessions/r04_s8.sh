#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_s8
mkdir -p "$OUT"
cd "$ROOT"
for mr in 2 4; do
  echo "== max run $mr" >> "$OUT/edge_bench.txt"
  timeout 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "runs_of_shared" 2>&1 | tail -1 >> "$OUT/edge_bench.txt"
  for g in n320_ico6 o96_ico5; do
    ANEMOI_AMD_EDGE_RUNS=1 python3 tools/edge_bench.py --set dec --graph $g $([ $g = o96_ico5 ] && echo --channels 512) >> "$OUT/edge_bench.txt" 2>&1
  done
done
ANEMOI_AMD_EDGE_RUNS=0 python3 tools/edge_bench.py --set dec --graph n320_ico6 >> "$OUT/edge_bench.txt" 2>&1
grep -v amdgpu "$OUT/edge_bench.txt"
