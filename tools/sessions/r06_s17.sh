#!/bin/bash
# round 6, session 14: the shelved kernel's LayerNorm-fold epilogue with its operands tied behind full waits (the compiler's counted wait + 0 / 1 / 2 / 8 wait states (variants 11, 12, 13, 10))
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_s17
mkdir -p "$OUT"
cd "$ROOT"
for v in 17 15; do
  echo "== variant ${v:-base}" | tee -a "$OUT/variants.txt"
  timeout 300 python3 tools/micro/run_with_lib.py tools/micro/bin/libanemoi_amd_kstream$v.so tools/micro/kstream_lnfold_repro.py --brief 2>&1 | grep "^M=" | cut -c1-230 | tee -a "$OUT/variants.txt"
done
