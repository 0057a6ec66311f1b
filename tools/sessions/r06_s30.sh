#!/bin/bash
# round 6, session 30: what distinguishes the slow class of boxes?  clocks / power caps next to a quick bench
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_s30_$$
mkdir -p "$OUT"
cd "$ROOT"
(rocm-smi --showclocks --showpower --showmaxpower --showperflevel --showtemp --showmemuse 2>&1 | head -60) > "$OUT/smi_idle.txt"
ANEMOI_AMD_GEMM_STAGGER=0,0,2 timeout 300 python3 bench.py --no-cpu-baseline --no-secondary > "$OUT/bench_off.json" 2>/dev/null &
sleep 25; (rocm-smi --showclocks --showpower --showtemp 2>&1 | head -40) > "$OUT/smi_busy.txt"; wait
echo "off: $(grep -o '"ms_per_step": [0-9.]*' "$OUT/bench_off.json" | head -1)"
timeout 300 python3 bench.py --no-cpu-baseline --no-secondary > "$OUT/bench_on.json" 2>/dev/null; echo "on: $(grep -o '"ms_per_step": [0-9.]*' "$OUT/bench_on.json" | head -1)"
grep -i "sclk\|mclk\|fclk\|power\|temp\|perf" "$OUT/smi_idle.txt" | head -20
echo ---busy; grep -i "sclk\|mclk\|fclk\|power\|temp" "$OUT/smi_busy.txt" | head -12
