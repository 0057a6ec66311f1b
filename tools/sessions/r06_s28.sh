#!/bin/bash
# round 6, session 28: what the residual / row-statistics epilogues cost on the mesh-sized GEMMs (isolated)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_s28
mkdir -p "$OUT"
cd "$ROOT"
GEMM_BENCH_BLASLT=0 GEMM_BENCH_STATS=1 timeout 300 python3 tools/gemm_bench.py 40962x1024x4096 40962x1024x1216 40962x4096x1024 40962x4288x1024 542080x1024x4096 542080x1024x1216 2>&1 | grep "^M=" | tee "$OUT/gemm_epilogues.txt"
