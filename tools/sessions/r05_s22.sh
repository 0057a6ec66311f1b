#!/bin/bash
# round 5, session 22: the last checkpointed region (decoder) keeps its activations -- tests, config-3 checkpointed step, peak memory
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s22
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python3 -m pytest tests/test_gpu_training.py tests/test_gpu_parity.py -q -m gpu -x -s -k "checkpoint or training_step or partitioned_training or graph or hierarchical" > "$OUT/pytest_ckpt.txt" 2>&1; grep -a "peak bytes\|passed\|failed" "$OUT/pytest_ckpt.txt" | tail -5
{
for last in 1 0; do
echo "== cfg3 checkpoint=1 ANEMOI_AMD_CHECKPOINT_LAST=$last"
ANEMOI_AMD_CHECKPOINT_LAST=$last ANEMOI_AMD_CHECKPOINT=1 python3 tools/train_step_bench.py cfg3 5
ANEMOI_AMD_CHECKPOINT_LAST=$last ANEMOI_AMD_CHECKPOINT=1 TRAIN_BENCH_GRAPH=1 python3 tools/train_step_bench.py cfg3 5
done
} > "$OUT/train_ckpt.txt" 2>&1; grep -v amdgpu "$OUT/train_ckpt.txt"
