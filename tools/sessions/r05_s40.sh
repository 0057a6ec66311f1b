#!/bin/bash
# round 5, session 40: the attention's run-to-run variation under CONTENTION (a GEMM loop and a twin process on the same GPU): shipped kernel and lab variants
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s40
mkdir -p "$OUT"
cd "$ROOT"
run_pair() {  # $1 = label, $2 = library ("" = shipped)
  ( while true; do timeout 60 python3 tools/gemm_bench.py > /dev/null 2>&1; done ) &
  NOISE=$!
  if [ -z "$2" ]; then
    timeout 200 python3 tools/micro/mhsa_repeat.py 800 40962 64 > "$OUT/$1.a.txt" 2>&1 &
    A=$!
    timeout 200 python3 tools/micro/mhsa_repeat.py 800 40962 64 > "$OUT/$1.b.txt" 2>&1
  else
    timeout 200 python3 tools/micro/run_with_lib.py "$2" tools/micro/mhsa_repeat.py 800 40962 64 > "$OUT/$1.a.txt" 2>&1 &
    A=$!
    timeout 200 python3 tools/micro/run_with_lib.py "$2" tools/micro/mhsa_repeat.py 800 40962 64 > "$OUT/$1.b.txt" 2>&1
  fi
  wait $A
  kill $NOISE 2>/dev/null; wait $NOISE 2>/dev/null
  echo "$1: $(tail -n 1 "$OUT/$1.a.txt" | cut -c1-80) | $(tail -n 1 "$OUT/$1.b.txt" | cut -c1-80)"
}
run_pair shipped ""
for v in 1 2 3 4; do run_pair lab$v anemoi_models_amd/lib/libanemoi_lab_att$v.so; done
run_pair shipped_again ""
