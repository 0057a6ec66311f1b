#!/bin/bash
# PMC passes of the fused Linear on one shape (plain epilogue): bash tools/gemm_pmc.sh 40962x4096x1024 [out-file]   (through gpurun)
# One rocprofv3 run per counter pair (SQ: 8 slots, but derived counters share them; TCC FETCH / WRITE separately), kernel-trace only.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
SHAPE=${1:-40962x4096x1024}
OUT=${2:-/dev/stdout}
cd /tmp && export TMPDIR=/tmp
{
echo "shape $SHAPE, tools/gemm_bench.py (3 warm-up + 10 timed launches of each epilogue; counters: mean per dispatch of the plain kernel)"
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM" "SQ_INSTS_SMEM SQ_INSTS_BRANCH" "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf /tmp/pmg
  GEMM_BENCH_BLASLT=0 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmg -- python3 $ROOT/tools/gemm_bench.py $SHAPE > /tmp/pmg.log 2>&1
  python3 $ROOT/tools/pmc_summary.py /tmp/pmg "w4_kernel<0, false, false" | sed 's/^.*linear_bf16_w4_kernel/  w4_kernel/'
done
rm -rf /tmp/pmg
GEMM_BENCH_BLASLT=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pmg -o kt -- python3 $ROOT/tools/gemm_bench.py $SHAPE > /tmp/pmg.log 2>&1
python3 $ROOT/tools/summarize_trace.py /tmp/pmg | head -8
grep "act=" /tmp/pmg.log
} > "$OUT" 2>&1
