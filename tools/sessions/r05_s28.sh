#!/bin/bash
# round 5, session 28: does the one-off failure of test_transformer_block_entry_point_is_the_op_by_op_route recur inside whole-file runs?
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s28
mkdir -p "$OUT"
cd "$ROOT"
for i in 1 2 3 4 5 6; do
  timeout 400 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x > "$OUT/run$i.txt" 2>&1
  grep -a "passed\|failed" "$OUT/run$i.txt" | tail -1
  grep -a "^FAILED" "$OUT/run$i.txt" | head -2
done
