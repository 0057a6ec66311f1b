#!/bin/bash
# round 6, session 32: cycle counters of the GEMM launches with the stagger off / on (raw per-kernel means kept)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_s32
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for i in 1 2; do for cfg in 0,0,2 2,16,2; do
  rm -rf /tmp/pmg
  ANEMOI_AMD_GEMM_STAGGER=$cfg rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d /tmp/pmg -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$OUT/pmc_${cfg}_$i.log" 2>&1
  echo "== stagger $cfg run $i: $(grep -o '"ms_per_step": [0-9.]*' "$OUT/pmc_${cfg}_$i.log" | head -1)"
  python3 "$ROOT/tools/pmc_summary.py" /tmp/pmg linear_bf16_w4 > "$OUT/pmc_${cfg}_$i.txt" 2>&1
done; done
