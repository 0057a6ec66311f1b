#!/bin/bash
# round 5, session 27: is test_transformer_block_entry_point_is_the_op_by_op_route[bf16-1024-16-2-700] flaky?  repeated runs + the rest of the suite
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s27
mkdir -p "$OUT"
cd "$ROOT"
for i in $(seq 1 12); do
  timeout 200 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "transformer_block_entry_point" 2>&1 | grep -a "passed\|failed" | tail -1
done > "$OUT/repeat.txt" 2>&1; sort "$OUT/repeat.txt" | uniq -c
timeout 300 python3 tools/determinism_check.py > "$OUT/determinism.txt" 2>&1; tail -5 "$OUT/determinism.txt"
timeout 1300 python3 -m pytest tests -q -m gpu --deselect "tests/test_gpu_parity.py::test_transformer_block_entry_point_is_the_op_by_op_route" > "$OUT/pytest_gpu.txt" 2>&1; grep -a "passed\|failed" "$OUT/pytest_gpu.txt" | tail -3
