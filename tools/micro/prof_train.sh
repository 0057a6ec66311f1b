# kernel summary of the config-3 training step (no checkpointing): bash tools/micro/prof_train.sh   (through gpurun)
cd /tmp && export TMPDIR=/tmp
export ANEMOI_AMD_CHECKPOINT=0
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/train_prof
rm -rf /tmp/ktb
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktb -o kt -- python3 $R/tools/train_step_bench.py cfg3 3 > $R/gpurun_out/train_prof/step.log 2>&1
python3 $R/tools/summarize_trace.py /tmp/ktb > $R/gpurun_out/train_prof/step_summary.txt 2>&1
grep "forward" $R/gpurun_out/train_prof/step.log
