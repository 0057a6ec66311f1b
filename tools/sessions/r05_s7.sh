#!/bin/bash
# round 5, session 7: whole GPU suite on the scheduled edge kernel; rank-of-N compute side; training-step kernel summaries
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s7
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python3 -m pytest tests -x -q -m gpu > "$OUT/pytest_gpu.txt" 2>&1; tail -4 "$OUT/pytest_gpu.txt"
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_cfg3.json" 2>/dev/null; python3 -c "
import json;d=json.load(open('$OUT/bench_cfg3.json'));print('cfg3', d['ms_per_step'], d['roofline']['frac'], d['roofline_edge']['frac'], d['roofline_edge']['avg_launch_ms'], d['kernel_time_ms'])"
ANEMOI_AMD_EDGE_SCHED=0 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_cfg3_nosched.json" 2>/dev/null; python3 -c "
import json;d=json.load(open('$OUT/bench_cfg3_nosched.json'));print('cfg3 no sched', d['ms_per_step'], d['roofline_edge']['frac'], d['roofline_edge']['avg_launch_ms'])"
python3 bench.py --workload cfg2 --steps 50 --warmup 10 --no-cpu-baseline > "$OUT/bench_cfg2.json" 2>/dev/null; python3 -c "
import json;d=json.load(open('$OUT/bench_cfg2.json'));print('cfg2', d['ms_per_step'])"
ANEMOI_AMD_EDGE_SCHED=0 python3 bench.py --workload cfg2 --steps 50 --warmup 10 --no-cpu-baseline > "$OUT/bench_cfg2_nosched.json" 2>/dev/null; python3 -c "
import json;d=json.load(open('$OUT/bench_cfg2_nosched.json'));print('cfg2 no sched', d['ms_per_step'])"
python3 tools/sim_rank.py --worlds 2,4,8 --steps 10 > "$OUT/sim_rank.txt" 2>&1; tail -12 "$OUT/sim_rank.txt"
ANEMOI_AMD_EDGE_SCHED=0 python3 tools/sim_rank.py --worlds 8 --steps 10 > "$OUT/sim_rank_nosched.txt" 2>&1; tail -4 "$OUT/sim_rank_nosched.txt"
bash tools/micro/prof_train.sh; cp gpurun_out/train_prof/step_summary.txt "$OUT/train_step_cfg3_summary.txt"; cp gpurun_out/train_prof/step.log "$OUT/train_step_cfg3.log"
bash tools/micro/prof_train_tfm.sh; cp gpurun_out/train_prof/tfm_summary.txt "$OUT/train_step_cfg3_transformer_summary.txt"; cp gpurun_out/train_prof/tfm.log "$OUT/train_step_cfg3_transformer.log"
