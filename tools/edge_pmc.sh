#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp

for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM" "SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" "GRBM_GUI_ACTIVE SQ_CYCLES" "SQ_IFETCH SQ_INSTS_BRANCH" "SQ_WAVES_EQ_64 SQ_LEVEL_WAVES" "TA_BUSY_avr TA_TA_BUSY_sum" "TCP_TA_TCP_STATE_READ_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum"; do
  rm -rf /tmp/pm
  rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pm -- python3 $ROOT/tools/edge_bench.py --iters 5 "$@" > /tmp/pm.log 2>&1
  python3 $ROOT/tools/pmc_summary.py /tmp/pm gt_edge | sed -E 's/^.*(folded|sched)_kernel[^ ]* *[a-z, 0-9]*, /  /'
done
