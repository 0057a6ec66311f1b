#!/bin/bash
# round 5, session 25: counters of the decoder launch on the group kernel (texture-address unit busy, instruction counts, fetch / write sizes)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s25
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for ctr in "GRBM_GUI_ACTIVE SQ_CYCLES" "TA_BUSY_avr TA_TA_BUSY_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "FETCH_SIZE WRITE_SIZE"; do
  rm -rf /tmp/pm
  rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pm -- python3 $ROOT/tools/edge_bench.py --iters 5 --set dec > /tmp/pm.log 2>&1
  python3 $ROOT/tools/pmc_summary.py /tmp/pm gt_edge
done > "$OUT/edge_pmc_dec_groups.txt" 2>&1
cat "$OUT/edge_pmc_dec_groups.txt" | cut -c1-60,100-200
