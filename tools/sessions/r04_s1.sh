#!/bin/bash
# Round 4, GPU session 1: 160-row tiles (MH = 5) + tile-height model.  Run: gpurun --timeout 1500 -- 'bash tools/sessions/r04_s1.sh'
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_s1
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "linear" > "$OUT/pytest_linear.txt" 2>&1
tail -3 "$OUT/pytest_linear.txt"
SH="5121x4096x1024 5121x1024x4096 5121x2048x1024 5121x2240x1024 5121x1024x1216 10242x2240x512 10242x2048x512 10242x512x2048 10242x512x704 5569x2048x1024"
timeout 300 python3 tools/gemm_bench.py $SH > "$OUT/gemm_default.txt" 2>&1
for mh in 4 5 6 8; do
  GEMM_BENCH_BLASLT=0 ANEMOI_AMD_GEMM_MH=$mh timeout 300 python3 tools/gemm_bench.py $SH > "$OUT/gemm_mh$mh.txt" 2>&1
done
timeout 600 python3 tools/sim_rank.py --worlds 4,8 --steps 10 > "$OUT/sim_rank.txt" 2>&1
timeout 300 python3 tools/sim_rank.py --worlds 8 --ranks 0 --steps 10 --detail > "$OUT/sim_rank8_detail.txt" 2>&1
timeout 300 python3 bench.py --workload cfg2 --steps 50 --warmup 10 --no-cpu-baseline --detail > "$OUT/bench_cfg2.txt" 2>&1
timeout 300 python3 bench.py --workload cfg2 --processor GNN --steps 20 --warmup 5 --no-cpu-baseline --detail > "$OUT/bench_cfg5.txt" 2>&1
timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_cfg3.txt" 2>&1
tail -2 "$OUT/sim_rank.txt"
