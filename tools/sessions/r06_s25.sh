#!/bin/bash
# round 6, session 25: final library -- the whole GPU suite, smoke(), the driver's default bench line (with its CPU baseline and the secondary block)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_s25
mkdir -p "$OUT"
cd "$ROOT"
SECONDS=0; timeout 1500 python3 -m pytest tests -m gpu -x -q --tb=short -rf > "$OUT/suite.txt" 2> "$OUT/suite.err"; echo "suite rc=$? ${SECONDS}s $(grep -E 'passed|failed' "$OUT/suite.txt" | tail -n 1)"; grep "^FAILED\|^E  " "$OUT/suite.txt" | cut -c1-300 | head -20
timeout 120 python3 __graft_entry__.py --smoke 2>&1 | tail -n 2
SECONDS=0; timeout 900 python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$? ${SECONDS}s $(cut -c1-400 "$OUT/bench_default.json")"
