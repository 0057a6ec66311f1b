#!/bin/bash
# round 6, session 5: which earlier test file moves the config-2 bf16 anchor inside the whole suite?
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_s5
mkdir -p "$OUT"
cd "$ROOT"
for f in test_abi test_bench_contract test_bounding test_distributed_cpu test_embed_fold test_gpu_attention_sizes; do
  timeout 900 python3 -m pytest tests/$f.py "tests/test_gpu_baseline_sizes.py::test_config2_bf16_anchored_to_the_oracle_under_bf16_autocast" -m gpu -q -s > "$OUT/$f.txt" 2>&1
  echo "$f rc=$? $(grep -o 'encoder latent: HIP [0-9.e-]*' "$OUT/$f.txt") $(tail -n 1 "$OUT/$f.txt")"
done
