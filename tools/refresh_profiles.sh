#!/bin/bash
# Regenerates the round's measurement evidence ON THE GPU BOX (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/refresh_profiles.sh r01'
# Writes gpurun_out/profiles_<tag>/ (small text/JSON/CSV only); tools/update_profiles.py then copies the
# summaries into profiles/ (tracked).  PMC passes are separate runs with --kernel-trace only (see MI355X_MICROARCH.md).
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp

cd "$ROOT"
python3 bench.py --steps 20 --warmup 5 > "$OUT/bench_cfg3_bf16.json" 2> "$OUT/bench_cfg3_bf16.err"
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --detail > "$OUT/bench_cfg3_bf16_detail.txt" 2>&1
python3 bench.py --workload cfg1 --steps 50 --warmup 10 --no-cpu-baseline > "$OUT/bench_cfg1_bf16.json" 2>/dev/null
python3 bench.py --workload cfg2 --steps 50 --warmup 10 --no-cpu-baseline > "$OUT/bench_cfg2_bf16.json" 2>/dev/null
python3 bench.py --dtype fp32 --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/bench_cfg3_fp32.json" 2>/dev/null
python3 bench.py --rollout 4 --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/bench_cfg4_rollout4_bf16.json" 2>/dev/null
python3 bench.py --workload cfg2 --processor GNN --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_cfg5_gnn_bf16.json" 2>/dev/null
python3 bench.py --processor Transformer --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_cfg3_transformer_bf16.json" 2>/dev/null
python3 bench.py --workload cfg2 --processor Transformer --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_cfg2_transformer_bf16.json" 2>/dev/null
python3 bench.py --workload cfg2 --steps 20 --warmup 5 > "$OUT/bench_cfg2_bf16_cpu_baseline.json" 2>/dev/null

cd /tmp
rm -rf /tmp/kt /tmp/pmcf /tmp/pmcw
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python3 "$ROOT/bench.py" --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > "$OUT/rocprof_bench.log" 2>&1
cp "$(find /tmp/kt -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv" 2>/dev/null
python3 "$ROOT/tools/summarize_trace.py" /tmp/kt > "$OUT/kernel_summary.txt" 2>&1

rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmcf -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python3 "$ROOT/tools/pmc_summary.py" /tmp/pmcf > "$OUT/pmc_fetch_size.txt" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmcw -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python3 "$ROOT/tools/pmc_summary.py" /tmp/pmcw > "$OUT/pmc_write_size.txt" 2>&1
# mesh-node self attention (Transformer processor): kernel trace + MFMA-busy counters of the attention kernel
rm -rf /tmp/ktt /tmp/pmcm
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktt -o kt -- python3 "$ROOT/bench.py" --processor Transformer --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 "$ROOT/tools/summarize_trace.py" /tmp/ktt > "$OUT/kernel_summary_transformer.txt" 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmcm -- python3 "$ROOT/bench.py" --processor Transformer --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 "$ROOT/tools/pmc_summary.py" /tmp/pmcm mhsa > "$OUT/pmc_mhsa_mfma_busy.txt" 2>&1
# round 4 additions: compute side of the 2 / 4 / 8-way partition, the training-step table, bench.py --gpus 8 with the ranks
# sharing this GPU (protocol transcript, not a measurement), kernel gaps of the rank-of-8 step
cd "$ROOT"
python3 tools/sim_rank.py --worlds 2,4,8 --steps 10 > "$OUT/sim_rank.txt" 2>&1
python3 tools/sim_rank.py --worlds 8 --ranks 0 --steps 10 --detail >> "$OUT/sim_rank.txt" 2>&1
bash tools/micro/train_bench_all.sh > "$OUT/train_step_bench.txt" 2>&1
TRAIN_BENCH_DROPOUT=0.1 python3 tools/train_step_bench.py cfg3 4 Transformer >> "$OUT/train_step_bench.txt" 2>&1
TRAIN_BENCH_DROPOUT=0.1 TRAIN_BENCH_GRAPH=1 python3 tools/train_step_bench.py cfg3 4 Transformer >> "$OUT/train_step_bench.txt" 2>&1
python3 bench.py --workload cfg2 --steps 50 --warmup 10 --no-cpu-baseline --hipgraph > "$OUT/bench_cfg2_bf16_hipgraph.json" 2>/dev/null
PORT=$((29500 + RANDOM % 400))
for r in 0 1 2 3 4 5 6 7; do
  ANEMOI_AMD_BENCH_SHARE_GPU=1 WORLD_SIZE=8 RANK=$r LOCAL_RANK=$r MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT \
    python3 bench.py --gpus 8 --workload cfg2 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_world8_shared_gpu.rank$r.txt" 2>&1 &
done
wait
cd /tmp
rm -rf /tmp/kt8
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt8 -o kt -- python3 "$ROOT/tools/sim_rank.py" --worlds 8 --ranks 0 --steps 6 > /dev/null 2>&1
python3 "$ROOT/tools/trace_gaps.py" /tmp/kt8 --last-frac 0.3 > "$OUT/gaps_rank8.txt" 2>&1
ls -la "$OUT"
