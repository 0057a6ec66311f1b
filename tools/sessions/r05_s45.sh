#!/bin/bash
# round 5, session 45: the contended run-to-run test, smoke() and the attention tests on the library as committed (lab variants removed, listings kept)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s45
mkdir -p "$OUT"
cd "$ROOT"
timeout 600 python3 -m pytest tests/test_gpu_attention_sizes.py -m gpu -x -q -k "second_process" > "$OUT/contended_test.txt" 2>&1
echo "contended test rc=$? $(tail -n 1 "$OUT/contended_test.txt")"
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/smoke.txt" 2>&1; tail -n 1 "$OUT/smoke.txt"
timeout 900 python3 -m pytest tests/test_gpu_attention_sizes.py tests/test_gpu_parity.py -m gpu -x -q -k "mhsa or attention or transformer or Transformer" > "$OUT/tests.txt" 2>&1
echo "tests rc=$? $(tail -n 1 "$OUT/tests.txt")"
