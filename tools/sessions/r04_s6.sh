#!/bin/bash
# Round 4, GPU session 6: decoder run kernel (parity + A/B timing), bf16 error values behind the 5e-2 bounds
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_s6
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "runs_of_shared or gt_edge or block_level or model_cfg1 or mapper" > "$OUT/pytest_edge.txt" 2>&1
tail -4 "$OUT/pytest_edge.txt"
for g in n320_ico6 o96_ico5; do
  ANEMOI_AMD_EDGE_RUNS=0 python3 tools/edge_bench.py --set dec --graph $g $([ $g = o96_ico5 ] && echo --channels 512) >> "$OUT/edge_bench.txt" 2>&1
  ANEMOI_AMD_EDGE_RUNS=1 python3 tools/edge_bench.py --set dec --graph $g $([ $g = o96_ico5 ] && echo --channels 512) >> "$OUT/edge_bench.txt" 2>&1
done
grep -v amdgpu "$OUT/edge_bench.txt"
ANEMOI_AMD_EDGE_RUNS=0 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_cfg3_noruns.txt" 2>&1
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --detail > "$OUT/bench_cfg3_runs.txt" 2>&1
grep -o '"ms_per_step": [0-9.]*' "$OUT/bench_cfg3_noruns.txt" "$OUT/bench_cfg3_runs.txt"
grep "edges=" "$OUT/bench_cfg3_runs.txt" | cut -c1-110
timeout 2400 python3 -m pytest tests/test_gpu_baseline_sizes.py tests/test_gpu_parity.py -q -m gpu -s -k "bf16 or config4 or hierarchical or interface_rollout or full_size" 2>&1 | grep -E "max rel|rel err|per-variable|passed|failed" > "$OUT/bf16_errors.txt"
cat "$OUT/bf16_errors.txt" | cut -c1-220
