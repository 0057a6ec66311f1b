#!/bin/bash
# round 6, session 9: A/B of two lab builds -- full wait states behind the eight-wave attention kernel's S^T products; staggered workgroup starts of the persistent GEMM
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_s9
mkdir -p "$OUT"
cd "$ROOT"
B=tools/micro/bin
timeout 300 python3 tools/micro/mhsa8_pad_ab.py /tmp/mhsa_ship.pt > "$OUT/mhsa8_ship.txt" 2>&1; cat "$OUT/mhsa8_ship.txt" | grep "S="
timeout 300 python3 tools/micro/run_with_lib.py $B/libanemoi_amd_mhsa8pad.so tools/micro/mhsa8_pad_ab.py /tmp/mhsa_pad.pt > "$OUT/mhsa8_pad.txt" 2>&1; cat "$OUT/mhsa8_pad.txt" | grep "S="
timeout 300 python3 tools/micro/mhsa8_pad_ab.py /tmp/mhsa_ship2.pt > "$OUT/mhsa8_ship2.txt" 2>&1; cat "$OUT/mhsa8_ship2.txt" | grep "S="
python3 tools/micro/mhsa8_pad_ab.py --compare /tmp/mhsa_ship.pt /tmp/mhsa_pad.pt | tee "$OUT/mhsa8_compare.txt"
SHAPES="542080x2048x256 542080x2240x256 542080x1024x256 40962x4096x1024 40962x1024x4096 40962x1024x1024 542080x4096x1024 542080x1024x4096 5121x4096x1024 5121x1024x4096 10242x2048x512"
for lib in ship stag2x1 stag2x2 stag3x1 stag4x1 ship; do
  echo "== $lib" | tee -a "$OUT/gemm_stagger.txt"
  if [ $lib = ship ]; then GEMM_BENCH_BLASLT=0 timeout 300 python3 tools/gemm_bench.py $SHAPES 2>&1 | grep "act=Identity res=False\|act=GELU" | tee -a "$OUT/gemm_stagger.txt"
  else GEMM_BENCH_BLASLT=0 timeout 300 python3 tools/gemm_bench.py --lib $B/libanemoi_amd_$lib.so $SHAPES 2>&1 | grep "act=Identity res=False\|act=GELU" | tee -a "$OUT/gemm_stagger.txt"; fi
done
for lib in ship stag2x1 stag4x1 ship; do
  if [ $lib = ship ]; then timeout 300 python3 bench.py --no-cpu-baseline --no-secondary > "$OUT/bench_$lib.json" 2>/dev/null
  else timeout 300 python3 tools/micro/bench_with_lib.py $B/libanemoi_amd_$lib.so --no-cpu-baseline --no-secondary > "$OUT/bench_$lib.json" 2>/dev/null; fi
  echo "bench $lib: $(grep -o '"ms_per_step": [0-9.]*' "$OUT/bench_$lib.json" | head -1) $(grep -o '"kernel_time_ms": {[^}]*}' "$OUT/bench_$lib.json")"
done
