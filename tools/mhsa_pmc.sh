#!/bin/bash
# PMC passes of the attention kernel (Transformer processor of config 3): bash tools/mhsa_pmc.sh   (through gpurun)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVES SQ_LEVEL_WAVES" "SQ_INSTS_BRANCH SQ_INSTS_SMEM" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES"; do
  rm -rf /tmp/pm
  rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pm -- python3 $ROOT/bench.py --processor Transformer --steps 1 --warmup 0 --no-cpu-baseline > /tmp/pm.log 2>&1
  python3 $ROOT/tools/pmc_summary.py /tmp/pm mhsa_bf16 | sed 's/^.*unsigned sho */  /'
done
