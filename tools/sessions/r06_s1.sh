#!/bin/bash
# round 6, session 1: the GPU suite with its duration table (what to move behind "gpu and slow"), the default bench line of this box
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_s1
mkdir -p "$OUT"
cd "$ROOT"
nproc > "$OUT/host.txt"; cat /sys/fs/cgroup/cpu.max >> "$OUT/host.txt" 2>&1; python3 -c "import torch,os; print(torch.get_num_threads(), len(os.sched_getaffinity(0)))" >> "$OUT/host.txt"
SECONDS=0; timeout 1500 python3 -m pytest tests -m gpu -q --durations=80 > "$OUT/suite.txt" 2> "$OUT/suite.err"; echo "suite rc=$? ${SECONDS}s $(tail -n 1 "$OUT/suite.txt")"
timeout 400 python3 bench.py --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "rc=$? $(tail -n 1 "$OUT/bench.json" | cut -c1-300)"
