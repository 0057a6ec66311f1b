#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s6
mkdir -p "$OUT"
cd "$ROOT"
python3 tools/micro/edge_sched_dbg.py > "$OUT/dbg.txt" 2>&1; grep -v amdgpu.ids "$OUT/dbg.txt"
cd /tmp && export TMPDIR=/tmp
for qb in 4 2; do
  export ANEMOI_AMD_MHSA_QB=$qb
  echo "== ANEMOI_AMD_MHSA_QB=$qb" >> "$OUT/mhsa_qb_pmc.txt"
  for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_WAVES"; do
    rm -rf /tmp/pm
    rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pm -- python3 $ROOT/tools/mhsa_bench.py > /tmp/pm.log 2>&1
    python3 $ROOT/tools/pmc_summary.py /tmp/pm mhsa_bf16_w4 | sed 's/^.*unsigned sho */  /' >> "$OUT/mhsa_qb_pmc.txt"
  done
  rm -rf /tmp/kt
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python3 $ROOT/tools/mhsa_bench.py > /tmp/kt.log 2>&1
  python3 $ROOT/tools/summarize_trace.py /tmp/kt 2>/dev/null | head -8 >> "$OUT/mhsa_qb_pmc.txt"
done
cat "$OUT/mhsa_qb_pmc.txt"
