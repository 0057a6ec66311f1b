#!/bin/bash
# round 5, session 17: attention backward with the tile loop unrolled by the LDS ring -- A/B against a lab build of the previous kernels, tests
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s17
mkdir -p "$OUT"
cd "$ROOT"
BASE=anemoi_models_amd/lib/libanemoi_lab_base.so
{
for rep in 1 2; do
echo "== base (ring position at run time)"
MHSA_BENCH_DROPOUT=0 python3 tools/micro/run_with_lib.py $BASE tools/mhsa_bwd_bench.py
MHSA_BENCH_DROPOUT=0.1 python3 tools/micro/run_with_lib.py $BASE tools/mhsa_bwd_bench.py
echo "== unrolled by the ring"
MHSA_BENCH_DROPOUT=0 python3 tools/mhsa_bwd_bench.py
MHSA_BENCH_DROPOUT=0.1 python3 tools/mhsa_bwd_bench.py
done
} > "$OUT/mhsa_ab.txt" 2>&1; grep -v amdgpu "$OUT/mhsa_ab.txt"
timeout 1500 python3 -m pytest tests/test_gpu_training.py tests/test_gpu_attention_sizes.py tests/test_gpu_parity.py -q -m gpu -k "dropout or mhsa or attention or graphed or transformer or Transformer" > "$OUT/pytest_att.txt" 2>&1; tail -4 "$OUT/pytest_att.txt"
