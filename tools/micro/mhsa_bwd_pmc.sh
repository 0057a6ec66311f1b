#!/bin/bash
# PMC passes of the two attention backward kernels at the config-3 shape: bash tools/micro/mhsa_bwd_pmc.sh   (through gpurun)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES"; do
  rm -rf /tmp/pm
  rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pm -- python3 $ROOT/tools/mhsa_bwd_bench.py 40962,16,64 > /tmp/pm.log 2>&1
  python3 $ROOT/tools/pmc_summary.py /tmp/pm mhsa_bwd | sed 's/(bf16 const.*AttnDropout) */ /; s/(bf16 const[^ ]* *[a-z,A-Z*0-9 ]*  */ /' | cut -c1-140
done
