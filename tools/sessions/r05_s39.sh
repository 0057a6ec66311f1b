#!/bin/bash
# round 5, session 39: does contention (a second process on the same GPU) bring out the attention's last-bit variation on a clean box?
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s39
mkdir -p "$OUT"
cd "$ROOT"
timeout 300 python3 tools/gemm_bench.py > "$OUT/noise_gemm.txt" 2>&1 &
NOISE=$!
timeout 300 python3 tools/micro/mhsa_repeat.py 3000 40962 64 > "$OUT/a.txt" 2>&1 &
A=$!
timeout 300 python3 tools/micro/mhsa_repeat.py 3000 40962 64 > "$OUT/b.txt" 2>&1
wait $A
kill $NOISE 2>/dev/null
tail -1 "$OUT/a.txt" "$OUT/b.txt"
grep -c iteration "$OUT/a.txt" "$OUT/b.txt"
