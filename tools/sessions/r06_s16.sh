#!/bin/bash
# round 6, session 16: reduced reproducer -- a consumer of a dwordx2 load's second dword as the very next instruction behind the wait
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_s16
mkdir -p "$OUT"
cd "$ROOT"
timeout 300 tools/micro/bin/vmcnt_consumer_race > "$OUT/race.txt" 2>&1; echo "rc=$?"; cat "$OUT/race.txt" | cut -c1-260
