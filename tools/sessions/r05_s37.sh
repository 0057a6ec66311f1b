#!/bin/bash
# round 5, session 37: does the attention's last-bit variation appear on a HOT chip?  a long run first, lab variants right behind it
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s37
mkdir -p "$OUT"
cd "$ROOT"
rocm-smi --showtemp --showclocks --showpower 2>/dev/null | grep -i "junction\|sclk\|Average Graphics" | head -4
timeout 400 python3 tools/micro/mhsa_repeat.py 16000 40962 64 > "$OUT/shipped_long.txt" 2>&1; tail -1 "$OUT/shipped_long.txt"; grep -c "iteration" "$OUT/shipped_long.txt"; grep "iteration" "$OUT/shipped_long.txt" | head -3 | cut -c1-100
rocm-smi --showtemp --showclocks --showpower 2>/dev/null | grep -i "junction\|sclk\|Average Graphics" | head -4
if grep -q " 0 of 16000" "$OUT/shipped_long.txt"; then echo "clean box, even hot"; exit 0; fi
for v in 1 2 3 4; do echo "== lab $v"; timeout 300 python3 tools/micro/run_with_lib.py anemoi_models_amd/lib/libanemoi_lab_att$v.so tools/micro/mhsa_repeat.py 6000 40962 64 2>&1 | tail -1; done
echo "== shipped again"; timeout 300 python3 tools/micro/mhsa_repeat.py 6000 40962 64 2>&1 | tail -1
