#!/bin/bash
# round 6, session 3: the bf16 anchor inside the suite's order (f32 forward of the same model object first)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_s3
mkdir -p "$OUT"
cd "$ROOT"
timeout 600 python3 tools/micro/latent_anchor_diag.py 16 > "$OUT/latent_diag.txt" 2>&1; echo "diag rc=$?"; grep threads "$OUT/latent_diag.txt" | cut -c1-300
timeout 600 python3 -m pytest tests/test_gpu_baseline_sizes.py -m gpu -q -s -k "config2" > "$OUT/config2.txt" 2>&1; echo "config2 rc=$?"; grep "config 2" "$OUT/config2.txt"; tail -n 2 "$OUT/config2.txt"
timeout 600 python3 -m pytest tests/test_gpu_baseline_sizes.py -m gpu -q -s -k "anchored" > "$OUT/anchored.txt" 2>&1; echo "anchored alone rc=$?"; grep "config 2" "$OUT/anchored.txt"; tail -n 2 "$OUT/anchored.txt"
