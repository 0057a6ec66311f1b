#!/bin/bash
# round 5, session 33: two more whole-suite runs (does the one-off failure recur?  the test now says which route moved)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s33
mkdir -p "$OUT"
cd "$ROOT"
for i in 1 2; do
  timeout 1300 python3 -m pytest tests -x -q -m gpu > "$OUT/run$i.txt" 2>&1
  grep -a "passed\|failed" "$OUT/run$i.txt" | tail -1
  grep -a "^FAILED\|block entry point" "$OUT/run$i.txt" | head -4
done
