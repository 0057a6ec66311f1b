#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s15
mkdir -p "$OUT"
cd "$ROOT"
python3 -c "import __graft_entry__ as g; g.build(); g.smoke()" > "$OUT/smoke.txt" 2>&1; tail -2 "$OUT/smoke.txt"
timeout 1200 python3 tools/fuzz_ops.py 150 4 > "$OUT/fuzz.txt" 2>&1; tail -6 "$OUT/fuzz.txt"
timeout 1200 python3 tools/determinism_check.py cfg2 20 > "$OUT/determinism.txt" 2>&1; tail -5 "$OUT/determinism.txt"
timeout 600 python3 tools/gemm_check.py > "$OUT/gemm_check.txt" 2>&1; tail -3 "$OUT/gemm_check.txt"
