#!/bin/bash
# round 5, session 1: the parity record (f32 16-block leg in the bench line, bf16 anchored to the oracle under bf16
# autocast, latent-error attribution) and bench.py's self-launch / watchdog / exchange timing
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s1
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python3 -m pytest tests/test_bench_contract.py -x -q -m gpu -s > "$OUT/bench_contract.txt" 2>&1; tail -5 "$OUT/bench_contract.txt"
timeout 900 python3 -m pytest tests/test_gpu_baseline_sizes.py -x -q -m gpu -s -k "config2" > "$OUT/config2.txt" 2>&1; tail -8 "$OUT/config2.txt"
timeout 900 python3 tools/latent_error.py --workload cfg3 > "$OUT/latent_error_cfg3.txt" 2>&1; tail -9 "$OUT/latent_error_cfg3.txt"
timeout 900 python3 tools/latent_error.py --workload cfg2 > "$OUT/latent_error_cfg2.txt" 2>&1; tail -9 "$OUT/latent_error_cfg2.txt"
timeout 1200 python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc $?"; tail -c 3000 "$OUT/bench_default.json"
