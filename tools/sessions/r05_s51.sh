#!/bin/bash
# round 5, session 51: bench.py with the run-to-run field (N = 1 default line without the CPU baseline, graph mode, the shared-GPU contract tests)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s51
mkdir -p "$OUT"
cd "$ROOT"
timeout 300 python3 bench.py --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "rc=$? $(tail -n 1 "$OUT/bench.json" | grep -o '"ms_per_step": [0-9.]*\|"run_to_run_identical": [a-z]*' | tr '\n' ' ')"
timeout 300 python3 bench.py --no-cpu-baseline --hipgraph > "$OUT/bench_graph.json" 2> "$OUT/bench_graph.err"; echo "rc=$? $(tail -n 1 "$OUT/bench_graph.json" | grep -o '"ms_per_step": [0-9.]*\|"run_to_run_identical": [a-z]*' | tr '\n' ' ')"
timeout 300 python3 bench.py --no-cpu-baseline --rollout 4 --steps 5 --warmup 2 > "$OUT/bench_r4.json" 2> "$OUT/bench_r4.err"; echo "rc=$? $(tail -n 1 "$OUT/bench_r4.json" | grep -o '"ms_per_step": [0-9.]*\|"run_to_run_identical": [a-z]*' | tr '\n' ' ')"
timeout 900 python3 -m pytest tests/test_bench_contract.py -m gpu -x -q > "$OUT/contract.txt" 2>&1; echo "contract rc=$? $(tail -n 1 "$OUT/contract.txt")"
