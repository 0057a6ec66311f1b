#!/bin/bash
# round 5, session 21: whole GPU suite on ABI 39, the bench line, the training-step table (GraphTransformer and Transformer)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s21
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python3 -m pytest tests -q -m gpu -x > "$OUT/pytest_gpu.txt" 2>&1; tail -3 "$OUT/pytest_gpu.txt"
python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; cut -c1-400 "$OUT/bench_default.json"
{
for ck in 0 1; do
echo "== cfg3 checkpoint=$ck"
ANEMOI_AMD_CHECKPOINT=$ck python3 tools/train_step_bench.py cfg3 5
ANEMOI_AMD_CHECKPOINT=$ck TRAIN_BENCH_GRAPH=1 python3 tools/train_step_bench.py cfg3 5
done
echo "== cfg2 checkpoint=0"
ANEMOI_AMD_CHECKPOINT=0 python3 tools/train_step_bench.py cfg2 10
ANEMOI_AMD_CHECKPOINT=0 TRAIN_BENCH_GRAPH=1 python3 tools/train_step_bench.py cfg2 10
echo "== cfg3 Transformer, checkpoint=0, no dropout / dropout 0.1"
ANEMOI_AMD_CHECKPOINT=0 python3 tools/train_step_bench.py cfg3 3 Transformer
ANEMOI_AMD_CHECKPOINT=0 TRAIN_BENCH_GRAPH=1 python3 tools/train_step_bench.py cfg3 3 Transformer
TRAIN_BENCH_DROPOUT=0.1 TRAIN_BENCH_GRAPH=1 python3 tools/train_step_bench.py cfg3 3 Transformer
} > "$OUT/train_table.txt" 2>&1; grep -v amdgpu "$OUT/train_table.txt"
