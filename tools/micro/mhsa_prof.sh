cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profiles_r03b
mkdir -p $OUT
cd $R
python3 bench.py --processor Transformer --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_cfg3_transformer_bf16.json 2>/dev/null
cd /tmp
rm -rf /tmp/ktt /tmp/pmcm
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktt -o kt -- python3 $R/bench.py --processor Transformer --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 $R/tools/summarize_trace.py /tmp/ktt > $OUT/kernel_summary_transformer.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmcm -- python3 $R/bench.py --processor Transformer --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 $R/tools/pmc_summary.py /tmp/pmcm mhsa > $OUT/pmc_mhsa_mfma_busy.txt 2>&1
cat $OUT/pmc_mhsa_mfma_busy.txt | grep w4
head -8 $OUT/kernel_summary_transformer.txt | cut -c1-60,100-170
