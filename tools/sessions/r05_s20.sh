#!/bin/bash
# round 5, session 20: training-step launch trimming (ragged rows inside the dual / actgrad launches, processor-wide
# parameter preparation, stacked gradient sinks) -- tests, then the config-3 training step A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s20
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_training.py tests/test_abi.py -q -m gpu -x -k "linear_dual or actgrad or batched_processor or training or checkpoint or graph or weight_grad or col_sum or edge or layer_norm or abi" > "$OUT/pytest_a.txt" 2>&1; tail -15 "$OUT/pytest_a.txt"
export ANEMOI_AMD_CHECKPOINT=0
{
for rep in 1 2; do
ANEMOI_AMD_TRAIN_BATCHED_PARAMS=0 python3 tools/train_step_bench.py cfg3 5
ANEMOI_AMD_TRAIN_BATCHED_PARAMS=1 python3 tools/train_step_bench.py cfg3 5
done
ANEMOI_AMD_TRAIN_BATCHED_PARAMS=1 TRAIN_BENCH_GRAPH=1 python3 tools/train_step_bench.py cfg3 5
ANEMOI_AMD_CHECKPOINT=1 ANEMOI_AMD_TRAIN_BATCHED_PARAMS=1 python3 tools/train_step_bench.py cfg3 5
} > "$OUT/train_ab.txt" 2>&1; grep -v amdgpu "$OUT/train_ab.txt"
