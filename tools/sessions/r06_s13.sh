#!/bin/bash
# round 6, session 13: the shelved K = 256 row-streaming GEMM's LayerNorm-fold epilogue -- where are the elements without the (-mean rstd) s[n] term?
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_s13
mkdir -p "$OUT"
cd "$ROOT"
timeout 600 python3 tools/micro/run_with_lib.py tools/micro/bin/libanemoi_amd_kstream.so tools/micro/kstream_lnfold_repro.py > "$OUT/repro.txt" 2>&1; echo "rc=$?"; grep -v amdgpu "$OUT/repro.txt" | cut -c1-260
