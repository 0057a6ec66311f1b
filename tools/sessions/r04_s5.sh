#!/bin/bash
# Round 4, GPU session 5: device-side dropout + backward grid + changed tests; GEMM PMC passes for the tile budget
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_s5
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python3 -m pytest tests/test_gpu_training.py -x -q -m gpu -k "dropout or mfma_route or graph" > "$OUT/pytest_training.txt" 2>&1
tail -5 "$OUT/pytest_training.txt"
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "block_level or mhsa or transformer" > "$OUT/pytest_parity.txt" 2>&1
tail -3 "$OUT/pytest_parity.txt"
timeout 900 python3 -m pytest tests/test_bench_contract.py tests/test_gpu_attention_sizes.py -x -q -m gpu > "$OUT/pytest_contract.txt" 2>&1
tail -3 "$OUT/pytest_contract.txt"
bash tools/gemm_pmc.sh 40962x4096x1024 "$OUT/gemm_pmc_40962x4096x1024.txt"
bash tools/gemm_pmc.sh 40962x4288x1024 "$OUT/gemm_pmc_40962x4288x1024.txt"
bash tools/gemm_pmc.sh 5121x4096x1024 "$OUT/gemm_pmc_5121x4096x1024.txt"
tail -12 "$OUT/gemm_pmc_40962x4096x1024.txt"
timeout 600 python3 tools/train_step_bench.py > "$OUT/train_step_bench.txt" 2>&1
tail -5 "$OUT/train_step_bench.txt"
