#!/bin/bash
# Round 4, GPU session 3: split-K (in-launch reduce-scatter) correctness + timing, bench contract, block ABI route
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_s3
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "linear or block_level or graphed" > "$OUT/pytest_linear.txt" 2>&1
tail -5 "$OUT/pytest_linear.txt"
SH="5121x1024x4096 5121x1024x1216 10242x512x2048 10242x512x704 5121x2048x1024"
for sk in -1 0 42 43 52 53 62 63 82 83; do
  if [ $sk = -1 ]; then unset ANEMOI_AMD_GEMM_SK; else export ANEMOI_AMD_GEMM_SK=$sk; fi
  GEMM_BENCH_BLASLT=0 GEMM_BENCH_STATS=1 timeout 300 python3 tools/gemm_bench.py $SH > "$OUT/gemm_sk$sk.txt" 2>&1
done
unset ANEMOI_AMD_GEMM_SK
timeout 1200 python3 -m pytest tests/test_bench_contract.py -x -q -m gpu > "$OUT/pytest_bench_contract.txt" 2>&1
tail -4 "$OUT/pytest_bench_contract.txt"
timeout 600 python3 tools/sim_rank.py --worlds 8 --steps 10 > "$OUT/sim_rank.txt" 2>&1
tail -2 "$OUT/sim_rank.txt"
timeout 300 python3 tools/sim_rank.py --worlds 8 --ranks 0 --steps 10 --detail > "$OUT/sim_rank8_detail.txt" 2>&1
timeout 300 python3 bench.py --workload cfg2 --steps 50 --warmup 10 --no-cpu-baseline --detail > "$OUT/bench_cfg2.txt" 2>&1
