#!/bin/bash
# round 5, session 49: default bench line and its kernel statistics on the final tree
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s49
mkdir -p "$OUT"
cd "$ROOT"
timeout 600 python3 bench.py > "$OUT/bench_cfg3.json" 2> "$OUT/bench_cfg3.err"; tail -n 1 "$OUT/bench_cfg3.json" | cut -c1-300
export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o cfg3 -- python3 "$ROOT/bench.py" --no-cpu-baseline --steps 10 --warmup 3 > "$OUT/prof.log" 2>&1
echo "rocprof rc=$?"
cd "$ROOT"
f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -n 1); if [ -n "$f" ]; then head -n 14 "$f" | cut -c1-220; fi
find "$OUT/prof" -name "*kernel_trace.csv" -size +30M -delete
