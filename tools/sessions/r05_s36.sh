#!/bin/bash
# round 5, session 36: the rare last-bit run-to-run variation of the D = 64 four-wave attention forward is box dependent -- look for a box that
# shows it, then compare lab variants of the tile ring's synchronisation ON THAT BOX
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s36
mkdir -p "$OUT"
cd "$ROOT"
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
timeout 300 python3 tools/micro/mhsa_repeat.py 4000 40962 64 > "$OUT/shipped.txt" 2>&1; tail -1 "$OUT/shipped.txt"
if grep -q " 0 of 4000" "$OUT/shipped.txt"; then echo "clean box"; exit 0; fi
for v in 1 2 3 4; do echo "== lab $v"; timeout 300 python3 tools/micro/run_with_lib.py anemoi_models_amd/lib/libanemoi_lab_att$v.so tools/micro/mhsa_repeat.py 4000 40962 64 2>&1 | tail -1; done
echo "== shipped again"; timeout 300 python3 tools/micro/mhsa_repeat.py 4000 40962 64 2>&1 | tail -1
