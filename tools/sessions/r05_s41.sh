#!/bin/bash
# round 5, session 41: which kernels vary run to run under contention (a GEMM loop + a twin process on the same GPU)?
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s41
mkdir -p "$OUT"
cd "$ROOT"
for what in linear mhsa8 edge ln; do
  ( while true; do timeout 60 python3 tools/gemm_bench.py > /dev/null 2>&1; done ) &
  NOISE=$!
  timeout 200 python3 tools/micro/ops_repeat.py $what 400 > "$OUT/$what.a.txt" 2>&1 &
  A=$!
  timeout 200 python3 tools/micro/ops_repeat.py $what 400 > "$OUT/$what.b.txt" 2>&1
  wait $A
  kill $NOISE 2>/dev/null; wait $NOISE 2>/dev/null
  echo "$(tail -n 1 "$OUT/$what.a.txt") | $(tail -n 1 "$OUT/$what.b.txt")"
done
echo "== alone"; for what in linear mhsa8; do timeout 200 python3 tools/micro/ops_repeat.py $what 300 2>&1 | tail -n 1; done
