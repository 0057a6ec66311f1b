#!/bin/bash
# round 5, session 43: the four-wave attention forward with the wait states of its prologue tied to the scores: run-to-run identity under contention, tests, speed
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s43
mkdir -p "$OUT"
cd "$ROOT"
( while true; do timeout 60 python3 tools/gemm_bench.py > /dev/null 2>&1; done ) &
NOISE=$!
timeout 300 python3 tools/micro/mhsa_repeat.py 1500 40962 64 > "$OUT/rep.a.txt" 2>&1 &
A=$!
timeout 300 python3 tools/micro/mhsa_repeat.py 1500 40962 64 > "$OUT/rep.b.txt" 2>&1
wait $A
timeout 200 python3 tools/micro/mhsa_repeat_diag.py 300 > "$OUT/diag.a.txt" 2>&1 &
A=$!
timeout 200 python3 tools/micro/mhsa_repeat_diag.py 300 > "$OUT/diag.b.txt" 2>&1
wait $A
timeout 200 python3 tools/micro/mhsa_repeat.py 600 700 64 > "$OUT/rep700.txt" 2>&1
kill $NOISE 2>/dev/null; wait $NOISE 2>/dev/null
echo "contention: $(tail -n 1 "$OUT/rep.a.txt" | cut -c1-90) | $(tail -n 1 "$OUT/rep.b.txt" | cut -c1-90)"
echo "diag: $(tail -n 1 "$OUT/diag.a.txt" | cut -c1-90) | $(tail -n 1 "$OUT/diag.b.txt" | cut -c1-90)"
echo "S=700: $(tail -n 1 "$OUT/rep700.txt" | cut -c1-90)"
sleep 3
timeout 900 python3 -m pytest tests/test_gpu_attention_sizes.py tests/test_gpu_parity.py -m gpu -x -q -k "mhsa or attention or transformer or Transformer" -W error::UserWarning > "$OUT/tests.txt" 2>&1
echo "tests rc=$? $(tail -n 1 "$OUT/tests.txt")"
timeout 300 python3 tools/mhsa_bench.py > "$OUT/mhsa_bench.txt" 2>&1; tail -n 6 "$OUT/mhsa_bench.txt" | cut -c1-200
