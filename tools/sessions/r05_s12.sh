#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s12
mkdir -p "$OUT"
cd "$ROOT"
{ for up in 16 12; do echo "== up $up"; DBG_UP=$up python3 tools/micro/edge_sched_dbg.py 2>&1 | grep -v amdgpu; done; } > "$OUT/dbg.txt" 2>&1; cat "$OUT/dbg.txt"
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "scheduled or folded or block_level or model" > "$OUT/pytest_a.txt" 2>&1; tail -4 "$OUT/pytest_a.txt"
ANEMOI_AMD_EDGE_SCHED=1 timeout 300 python3 tools/edge_bench.py --set proc --iters 50 2>&1 | grep -v amdgpu
