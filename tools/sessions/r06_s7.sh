#!/bin/bash
# round 6, session 7: first forwards on poisoned free lists (what a test sees in mid-suite), the invariants test that failed inside the suite, PMC of the LDS-tile edge kernel
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_s7
mkdir -p "$OUT"
cd "$ROOT"
timeout 1200 python3 tools/micro/poison_forward.py > "$OUT/poison.txt" 2>&1; echo "poison rc=$?"; grep "poisoned\|Error" "$OUT/poison.txt" | cut -c1-330
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "full_size_invariants_n320" > "$OUT/invariants.txt" 2>&1; echo "invariants rc=$? $(tail -n 1 "$OUT/invariants.txt")"; grep "^E " "$OUT/invariants.txt" | head -5
cd /tmp && export TMPDIR=/tmp
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "TA_TA_BUSY_sum TA_BUSY_avr" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum"; do
  rm -rf /tmp/pm
  ANEMOI_AMD_EDGE_TILES=1 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pm -- python3 $ROOT/tools/edge_bench.py --iters 5 > /tmp/pm.log 2>&1
  python3 $ROOT/tools/pmc_summary.py /tmp/pm gt_edge | cut -c1-200 >> "$OUT/tiles_pmc.txt"
done
cat "$OUT/tiles_pmc.txt" | sed -E 's/^.*(folded|sched|tiles)_kernel[^ ]* */  /' | cut -c1-160
