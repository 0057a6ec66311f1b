#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s10
mkdir -p "$OUT"
cd "$ROOT"
{
echo "== up 16 U 3"; DBG_UP=16 ANEMOI_AMD_EDGE_U=3 python3 tools/micro/edge_sched_dbg.py 2>&1 | grep -v amdgpu
echo "== up 16 U 6"; DBG_UP=16 ANEMOI_AMD_EDGE_U=6 python3 tools/micro/edge_sched_dbg.py 2>&1 | grep -v amdgpu
echo "== up 8"; DBG_UP=8 python3 tools/micro/edge_sched_dbg.py 2>&1 | grep -v amdgpu
echo "== up 4"; DBG_UP=4 python3 tools/micro/edge_sched_dbg.py 2>&1 | grep -v amdgpu
echo "== up 16 zero u"; DBG_UP=16 DBG_ZERO_U=1 python3 tools/micro/edge_sched_dbg.py 2>&1 | grep -v amdgpu
echo "== up 16 zero attr"; DBG_UP=16 DBG_ZERO_ATTR=1 python3 tools/micro/edge_sched_dbg.py 2>&1 | grep -v amdgpu
} > "$OUT/dbg.txt" 2>&1
cat "$OUT/dbg.txt"
