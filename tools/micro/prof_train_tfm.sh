# kernel summary of the config-3 training step of the Transformer-processor model: bash tools/micro/prof_train_tfm.sh   (through gpurun)
cd /tmp && export TMPDIR=/tmp
export ANEMOI_AMD_CHECKPOINT=0
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/train_prof
rm -rf /tmp/ktt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktt -o kt -- python3 $R/tools/train_step_bench.py cfg3 2 Transformer > $R/gpurun_out/train_prof/tfm.log 2>&1
python3 $R/tools/summarize_trace.py /tmp/ktt > $R/gpurun_out/train_prof/tfm_summary.txt 2>&1
grep "forward" $R/gpurun_out/train_prof/tfm.log
