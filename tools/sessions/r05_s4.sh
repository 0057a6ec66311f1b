#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s4
mkdir -p "$OUT"
cd "$ROOT"
python3 tools/micro/edge_sched_dbg.py > "$OUT/dbg.txt" 2>&1; grep -v amdgpu.ids "$OUT/dbg.txt"
{
for w in 3 4 5; do
ANEMOI_AMD_EDGE_WGS=$w timeout 300 python3 tools/edge_bench.py --set proc --iters 50
done
ANEMOI_AMD_EDGE_SCHED=0 timeout 300 python3 tools/edge_bench.py --set proc --iters 50
for w in 3 4 5; do
ANEMOI_AMD_EDGE_WGS=$w timeout 300 python3 tools/edge_bench.py --graph o96_ico5 --channels 512 --set proc --iters 50
done
ANEMOI_AMD_EDGE_SCHED=0 timeout 300 python3 tools/edge_bench.py --graph o96_ico5 --channels 512 --set proc --iters 50
} > "$OUT/edge_ab.txt" 2>&1
grep -v amdgpu.ids "$OUT/edge_ab.txt"
