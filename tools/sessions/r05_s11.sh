#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s11
mkdir -p "$OUT"
cd "$ROOT"
{ for up in 16 12 8 4; do echo "== up $up"; DBG_UP=$up python3 tools/micro/edge_sched_dbg.py 2>&1 | grep -v amdgpu; done; } > "$OUT/dbg.txt" 2>&1; cat "$OUT/dbg.txt"
timeout 1700 python3 -m pytest tests -q -m gpu > "$OUT/pytest_gpu.txt" 2>&1; tail -6 "$OUT/pytest_gpu.txt"
