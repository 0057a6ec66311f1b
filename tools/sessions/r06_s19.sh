#!/bin/bash
# round 6, session 19: staggered start of the four-wave attention kernel (lab env), and the GEMM stagger default on the Transformer model
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_s19
mkdir -p "$OUT"
cd "$ROOT"
for i in 1 2; do for u in 0 2 4 8 16 32; do
  echo "mhsa stagger unit $u run $i: $(ANEMOI_AMD_MHSA_STAGGER=$u timeout 120 python3 tools/mhsa_bench.py 2>/dev/null | head -n 1)" | tee -a "$OUT/mhsa_stagger.txt"
done; done
