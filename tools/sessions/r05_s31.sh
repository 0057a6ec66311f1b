#!/bin/bash
# round 5, session 31: HBM traffic of the final library's kernels (FETCH_SIZE and WRITE_SIZE in separate passes, each under a timeout)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_r05t
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcf /tmp/pmcw
timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmcf -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; echo "fetch pass rc $?"
timeout 120 python3 "$ROOT/tools/pmc_summary.py" /tmp/pmcf > "$OUT/pmc_fetch_size.txt" 2>&1
timeout 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmcw -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; echo "write pass rc $?"
timeout 120 python3 "$ROOT/tools/pmc_summary.py" /tmp/pmcw > "$OUT/pmc_write_size.txt" 2>&1
wc -l "$OUT"/*.txt
