#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s9
mkdir -p "$OUT"
cd "$ROOT"
DBG_UP=16 python3 tools/micro/edge_sched_dbg.py > "$OUT/dbg16.txt" 2>&1; grep -v amdgpu.ids "$OUT/dbg16.txt"
DBG_UP=12 python3 tools/micro/edge_sched_dbg.py > "$OUT/dbg12.txt" 2>&1; grep -v amdgpu.ids "$OUT/dbg12.txt"
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "scheduled or runs_of_shared or node_partitioned_forward" > "$OUT/pytest_a.txt" 2>&1; tail -4 "$OUT/pytest_a.txt"
timeout 900 python3 -m pytest tests/test_gpu_training.py -q -m gpu -k "node_partitioned_training" > "$OUT/pytest_b.txt" 2>&1; tail -4 "$OUT/pytest_b.txt"
{
ANEMOI_AMD_EDGE_RUNS=1 timeout 300 python3 tools/edge_bench.py --set dec --iters 30
ANEMOI_AMD_EDGE_RUNS=1 ANEMOI_AMD_EDGE_RUNS_WGS=5 timeout 300 python3 tools/edge_bench.py --set dec --iters 30
ANEMOI_AMD_EDGE_RUNS=1 ANEMOI_AMD_EDGE_RUNS_WGS=3 timeout 300 python3 tools/edge_bench.py --set dec --iters 30
ANEMOI_AMD_EDGE_RUNS=0 ANEMOI_AMD_EDGE_SCHED=0 timeout 300 python3 tools/edge_bench.py --set dec --iters 30
} > "$OUT/edge_dec.txt" 2>&1
grep -v amdgpu.ids "$OUT/edge_dec.txt"
