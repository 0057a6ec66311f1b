#!/bin/bash
# round 5, session 18: full kernel list of the config-3 GraphTransformer training step on the current library
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s18
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export ANEMOI_AMD_CHECKPOINT=0
python3 $ROOT/tools/train_step_bench.py cfg3 5 > "$OUT/step.txt" 2>&1; grep -v amdgpu "$OUT/step.txt"
rm -rf /tmp/ktb
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktb -o kt -- python3 $ROOT/tools/train_step_bench.py cfg3 3 > "$OUT/step_prof.log" 2>&1
python3 $ROOT/tools/summarize_trace.py /tmp/ktb 0 > "$OUT/step_summary_full.txt" 2>&1
rm -rf /tmp/ktf
TRAIN_BENCH_PHASE=forward rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktf -o kt -- python3 $ROOT/tools/train_step_bench.py cfg3 3 > "$OUT/fwd_prof.log" 2>&1
python3 $ROOT/tools/summarize_trace.py /tmp/ktf 0 > "$OUT/fwd_summary_full.txt" 2>&1
tail -1 "$OUT/step_summary_full.txt" "$OUT/fwd_summary_full.txt"
