#!/bin/bash
# round 5, session 26: verification of the round's final library (ABI 39 + group kernel): whole GPU suite, bench lines, kernel summary,
# compute side of the partitions, bench.py --gpus 8 with the ranks sharing this GPU.  Every command under its own timeout.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s26
mkdir -p "$OUT"
cd "$ROOT"
timeout 1300 python3 -m pytest tests -q -m gpu -x > "$OUT/pytest_gpu.txt" 2>&1; grep -a "passed\|failed" "$OUT/pytest_gpu.txt" | tail -2
timeout 600 python3 bench.py --steps 20 --warmup 5 > "$OUT/bench_cfg3_bf16.json" 2> "$OUT/bench_cfg3_bf16.err"; cut -c1-330 "$OUT/bench_cfg3_bf16.json"
timeout 300 python3 bench.py --workload cfg2 --steps 50 --warmup 10 --no-cpu-baseline > "$OUT/bench_cfg2_bf16.json" 2>/dev/null; cut -c1-200 "$OUT/bench_cfg2_bf16.json"
timeout 300 python3 bench.py --workload cfg2 --processor GNN --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_cfg5_gnn_bf16.json" 2>/dev/null; cut -c1-200 "$OUT/bench_cfg5_gnn_bf16.json"
timeout 300 python3 bench.py --processor Transformer --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_cfg3_transformer_bf16.json" 2>/dev/null; cut -c1-200 "$OUT/bench_cfg3_transformer_bf16.json"
timeout 300 python3 bench.py --rollout 4 --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/bench_cfg4_rollout4_bf16.json" 2>/dev/null; cut -c1-200 "$OUT/bench_cfg4_rollout4_bf16.json"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python3 "$ROOT/bench.py" --steps 8 --warmup 2 --no-cpu-baseline > "$OUT/rocprof_bench.log" 2>&1
cp "$(find /tmp/kt -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv" 2>/dev/null
timeout 120 python3 "$ROOT/tools/summarize_trace.py" /tmp/kt > "$OUT/kernel_summary.txt" 2>&1; head -12 "$OUT/kernel_summary.txt" | cut -c1-60,100-170
cd "$ROOT"
timeout 500 python3 tools/sim_rank.py --worlds 2,4,8 --steps 10 > "$OUT/sim_rank.txt" 2>&1; grep -v amdgpu "$OUT/sim_rank.txt" | tail -8
ANEMOI_AMD_BENCH_SHARE_GPU=1 timeout 900 python3 bench.py --gpus 8 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_world8_cfg3_shared_gpu.txt" 2>&1; cut -c1-400 "$OUT/bench_world8_cfg3_shared_gpu.txt" | tail -3
