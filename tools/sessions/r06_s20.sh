#!/bin/bash
# round 6, session 20: the whole GPU suite on the final library (ABI 40, staggered GEMM start, tile kernel off by default), then the round's profile refresh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_s20
mkdir -p "$OUT"
cd "$ROOT"
SECONDS=0; timeout 1500 python3 -m pytest tests -m gpu -q --tb=short -rf --durations=25 > "$OUT/suite.txt" 2> "$OUT/suite.err"; echo "suite rc=$? ${SECONDS}s $(grep -E 'passed|failed' "$OUT/suite.txt" | tail -n 1)"; grep "^FAILED\|^E  " "$OUT/suite.txt" | cut -c1-300 | head -20
timeout 120 python3 __graft_entry__.py --smoke 2>&1 | tail -n 2
bash tools/refresh_profiles.sh r06 > "$OUT/refresh.log" 2>&1; echo "refresh rc=$?"; tail -n 3 "$OUT/refresh.log"
