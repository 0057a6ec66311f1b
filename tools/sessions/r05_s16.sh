#!/bin/bash
# round 5, session 16: the cheaper attention-dropout mask generator -- tests, kernel times, training step
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s16
mkdir -p "$OUT"
cd "$ROOT"
timeout 1200 python3 -m pytest tests/test_gpu_training.py tests/test_gpu_attention_sizes.py -q -m gpu -k "dropout or mhsa or attention or graphed or transformer or Transformer" > "$OUT/pytest_dropout.txt" 2>&1; tail -4 "$OUT/pytest_dropout.txt"
{
MHSA_BENCH_DROPOUT=0 python3 tools/mhsa_bwd_bench.py
MHSA_BENCH_DROPOUT=0.1 python3 tools/mhsa_bwd_bench.py
} > "$OUT/mhsa_dropout.txt" 2>&1; grep -v amdgpu "$OUT/mhsa_dropout.txt"
{
TRAIN_BENCH_DROPOUT=0.1 python3 tools/train_step_bench.py cfg3 3 Transformer
TRAIN_BENCH_DROPOUT=0.1 TRAIN_BENCH_GRAPH=1 python3 tools/train_step_bench.py cfg3 3 Transformer
} > "$OUT/train_tfm_dropout.txt" 2>&1; grep -v amdgpu "$OUT/train_tfm_dropout.txt"
