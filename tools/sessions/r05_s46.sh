#!/bin/bash
# round 5, session 46: whole GPU suite on the final tree (exact run-to-run checks + the contended test), kernel trace of the Transformer line after the attention fix
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s46
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python3 -m pytest tests -m gpu -x -q > "$OUT/suite.txt" 2>&1
echo "suite rc=$? $(grep -E "passed|failed" "$OUT/suite.txt" | tail -n 1)"
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_tfm" -o tfm -- python3 "$ROOT/bench.py" --processor Transformer --no-cpu-baseline --steps 5 --warmup 2 > "$OUT/prof_tfm.log" 2>&1
echo "rocprof rc=$?"; tail -n 1 "$OUT/prof_tfm.log" | cut -c1-200
cd "$ROOT"
f=$(find "$OUT/prof_tfm" -name "*kernel_stats.csv" | head -n 1); [ -n "$f" ] && head -n 12 "$f" | cut -c1-200
