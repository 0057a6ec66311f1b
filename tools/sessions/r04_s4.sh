#!/bin/bash
# Round 4, GPU session 4: kernel traces (gaps) of the rank-of-8 step, config 2, config 3; then the whole GPU suite
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_s4
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt8 /tmp/kt2 /tmp/kt3
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt8 -o kt -- python3 "$ROOT/tools/sim_rank.py" --worlds 8 --ranks 0 --steps 6 > "$OUT/sim8.log" 2>&1
python3 "$ROOT/tools/trace_gaps.py" /tmp/kt8 --last-frac 0.3 > "$OUT/gaps_rank8.txt" 2>&1
python3 "$ROOT/tools/summarize_trace.py" /tmp/kt8 > "$OUT/kernels_rank8.txt" 2>&1
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt2 -o kt -- python3 "$ROOT/bench.py" --workload cfg2 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/cfg2.log" 2>&1
python3 "$ROOT/tools/trace_gaps.py" /tmp/kt2 --last-frac 0.5 > "$OUT/gaps_cfg2.txt" 2>&1
python3 "$ROOT/tools/summarize_trace.py" /tmp/kt2 > "$OUT/kernels_cfg2.txt" 2>&1
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt3 -o kt -- python3 "$ROOT/bench.py" --steps 8 --warmup 2 --no-cpu-baseline > "$OUT/cfg3.log" 2>&1
python3 "$ROOT/tools/trace_gaps.py" /tmp/kt3 --last-frac 0.5 > "$OUT/gaps_cfg3.txt" 2>&1
cd "$ROOT"
python3 bench.py --workload cfg2 --steps 50 --warmup 10 --no-cpu-baseline --hipgraph > "$OUT/bench_cfg2_hipgraph.txt" 2>&1
head -3 "$OUT/gaps_rank8.txt"; head -3 "$OUT/gaps_cfg2.txt"; head -3 "$OUT/gaps_cfg3.txt"
timeout 3000 python3 -m pytest tests -x -q -m gpu > "$OUT/pytest_gpu.txt" 2>&1
tail -5 "$OUT/pytest_gpu.txt"
