# kernel summary of the DIFFERENTIABLE forward of config 3 alone (no checkpointing): bash tools/micro/prof_train_fwd.sh   (through gpurun)
cd /tmp && export TMPDIR=/tmp
export ANEMOI_AMD_CHECKPOINT=0
export TRAIN_BENCH_PHASE=forward
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/train_prof
rm -rf /tmp/ktf
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktf -o kt -- python3 $R/tools/train_step_bench.py cfg3 3 > $R/gpurun_out/train_prof/fwd.log 2>&1
python3 $R/tools/summarize_trace.py /tmp/ktf > $R/gpurun_out/train_prof/fwd_summary.txt 2>&1
tail -3 $R/gpurun_out/train_prof/fwd.log
