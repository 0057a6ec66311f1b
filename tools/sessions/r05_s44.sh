#!/bin/bash
# round 5, session 44: whole GPU suite with the run-to-run checks exact again, clean attention / Transformer / default bench lines,
# and the formerly flaky tests once more under contention (a GEMM loop on the same GPU)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s44
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python3 -m pytest tests -m gpu -x -q > "$OUT/suite.txt" 2>&1
echo "suite rc=$? $(tail -n 1 "$OUT/suite.txt")"
timeout 300 python3 tools/mhsa_bench.py > "$OUT/mhsa_bench.txt" 2>&1; grep "TFLOP" "$OUT/mhsa_bench.txt" | cut -c1-200
timeout 600 python3 bench.py --processor Transformer --no-cpu-baseline > "$OUT/bench_tfm.json" 2> "$OUT/bench_tfm.err"; tail -n 1 "$OUT/bench_tfm.json" | cut -c1-400
timeout 900 python3 bench.py > "$OUT/bench_cfg3.json" 2> "$OUT/bench_cfg3.err"; tail -n 1 "$OUT/bench_cfg3.json" | cut -c1-600
( while true; do timeout 45 python3 tools/gemm_bench.py > /dev/null 2>&1; done ) &
NOISE=$!
timeout 600 python3 -m pytest tests/test_gpu_attention_sizes.py tests/test_gpu_parity.py tests/test_gpu_training.py -m gpu -q -k "repeated or transformer_block_entry or mesh_size or dropout" > "$OUT/contended.txt" 2>&1
echo "contended rc=$? $(tail -n 1 "$OUT/contended.txt")"
kill $NOISE 2>/dev/null; wait $NOISE 2>/dev/null
sleep 50
