#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_s7
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "runs_of_shared or block_level" > "$OUT/pytest_edge.txt" 2>&1
tail -3 "$OUT/pytest_edge.txt"
for g in n320_ico6 o96_ico5; do
  ANEMOI_AMD_EDGE_RUNS=0 python3 tools/edge_bench.py --set dec --graph $g $([ $g = o96_ico5 ] && echo --channels 512) >> "$OUT/edge_bench.txt" 2>&1
  ANEMOI_AMD_EDGE_RUNS=1 python3 tools/edge_bench.py --set dec --graph $g $([ $g = o96_ico5 ] && echo --channels 512) >> "$OUT/edge_bench.txt" 2>&1
done
grep -v amdgpu "$OUT/edge_bench.txt"
