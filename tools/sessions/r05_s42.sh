#!/bin/bash
# round 5, session 42: WHERE two runs of the four-wave attention forward differ under contention
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s42
mkdir -p "$OUT"
cd "$ROOT"
( while true; do timeout 60 python3 tools/gemm_bench.py > /dev/null 2>&1; done ) &
NOISE=$!
timeout 200 python3 tools/micro/mhsa_repeat_diag.py 300 > "$OUT/a.txt" 2>&1 &
A=$!
timeout 200 python3 tools/micro/mhsa_repeat_diag.py 300 > "$OUT/b.txt" 2>&1
wait $A
kill $NOISE 2>/dev/null; wait $NOISE 2>/dev/null
cut -c1-900 "$OUT/a.txt" | tail -n 8
