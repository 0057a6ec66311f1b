#!/bin/bash
# round 5, session 34: training input assembled once in the GEMM layout -- tests, config-3 / config-2 training step A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s34
mkdir -p "$OUT"
cd "$ROOT"
timeout 1200 python3 -m pytest tests/test_gpu_training.py tests/test_gpu_parity.py tests/test_gpu_baseline_sizes.py -q -m gpu -x -k "wide_dx or training or checkpoint or graph or hierarchical or batched or bf16" > "$OUT/pytest.txt" 2>&1; grep -a "passed\|failed" "$OUT/pytest.txt" | tail -2; grep -a "^FAILED\|Error" "$OUT/pytest.txt" | head -5
export ANEMOI_AMD_CHECKPOINT=0
{
for rep in 1 2; do
echo "== cat / cast input"; ANEMOI_AMD_TRAIN_ASSEMBLE=0 timeout 300 python3 tools/train_step_bench.py cfg3 5
echo "== assembled input"; timeout 300 python3 tools/train_step_bench.py cfg3 5
done
echo "== as one graph"; TRAIN_BENCH_GRAPH=1 timeout 300 python3 tools/train_step_bench.py cfg3 5
echo "== cfg2"; ANEMOI_AMD_TRAIN_ASSEMBLE=0 TRAIN_BENCH_GRAPH=1 timeout 300 python3 tools/train_step_bench.py cfg2 10; TRAIN_BENCH_GRAPH=1 timeout 300 python3 tools/train_step_bench.py cfg2 10
} > "$OUT/train_ab.txt" 2>&1; grep -v amdgpu "$OUT/train_ab.txt"
timeout 300 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | cut -c1-200
