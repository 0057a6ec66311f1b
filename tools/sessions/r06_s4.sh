#!/bin/bash
# round 6, session 4: uninitialised reads?  poisoned allocator free lists; the suite's order around the moved anchor
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_s4
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python3 tools/micro/poison_forward.py > "$OUT/poison.txt" 2>&1; echo "poison rc=$?"; grep "after poisoning\|Error" "$OUT/poison.txt" | cut -c1-250
timeout 900 python3 -m pytest tests/test_bench_contract.py tests/test_gpu_baseline_sizes.py -m gpu -q -s -k "secondary or anchored" > "$OUT/order.txt" 2>&1; echo "order rc=$?"; grep "config 2" "$OUT/order.txt"; tail -n 2 "$OUT/order.txt"
