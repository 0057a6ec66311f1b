#!/bin/bash
# round 5, session 23: decoder edge kernel on groups of destinations that share their three sources -- tests, A/B, bench
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s23
mkdir -p "$OUT"
cd "$ROOT"
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_abi.py tests/test_gpu_baseline_sizes.py -q -m gpu -x -k "runs or edge or folded or abi or mapper or config2 or block" > "$OUT/pytest_groups.txt" 2>&1; grep -a "passed\|failed" "$OUT/pytest_groups.txt" | tail -3
{
for rep in 1 2; do
echo "== runs (consecutive, <= 2)"; ANEMOI_AMD_EDGE_GROUPS=0 python3 tools/edge_bench.py --set dec --iters 30
echo "== groups (source triple, <= 8)"; python3 tools/edge_bench.py --set dec --iters 30
done
for w in 3 5 6; do echo "== groups, $w workgroups per CU"; ANEMOI_AMD_EDGE_GROUP_WGS=$w python3 tools/edge_bench.py --set dec --iters 30; done
echo "== plain"; ANEMOI_AMD_EDGE_RUNS=0 python3 tools/edge_bench.py --set dec --iters 30
} > "$OUT/edge_dec_ab.txt" 2>&1; grep -v amdgpu "$OUT/edge_dec_ab.txt"
python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; cut -c1-260 "$OUT/bench_default.json"
ANEMOI_AMD_EDGE_GROUPS=0 python3 bench.py > "$OUT/bench_runs.json" 2> "$OUT/bench_runs.err"; cut -c1-260 "$OUT/bench_runs.json"
