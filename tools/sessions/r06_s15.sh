#!/bin/bash
# round 6, session 15: the shelved kernel's LayerNorm-fold epilogue -- the compiler's counted wait + 0 / 1 / 2 / 8 wait states, statistics tied (variants 11, 12, 13, 10)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_s15
mkdir -p "$OUT"
cd "$ROOT"
for v in 11 12 13 10; do
  echo "== variant ${v:-base}" | tee -a "$OUT/variants.txt"
  timeout 300 python3 tools/micro/run_with_lib.py tools/micro/bin/libanemoi_amd_kstream$v.so tools/micro/kstream_lnfold_repro.py --brief 2>&1 | grep "^M=" | cut -c1-230 | tee -a "$OUT/variants.txt"
done
