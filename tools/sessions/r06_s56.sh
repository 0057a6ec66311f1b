#!/bin/bash
# round 6, session 56: the small-configuration bench lines again with the final bench.py (instrumented forward behind the stream hold)
set -u
OUT=gpurun_out/r06_s56; mkdir -p $OUT
python3 bench.py --workload cfg1 --steps 50 --warmup 10 --no-cpu-baseline > "$OUT/bench_cfg1_bf16.json" 2>/dev/null
python3 bench.py --workload cfg2 --steps 50 --warmup 10 --no-cpu-baseline > "$OUT/bench_cfg2_bf16.json" 2>/dev/null
python3 bench.py --workload cfg2 --processor GNN --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_cfg5_gnn_bf16.json" 2>/dev/null
python3 bench.py --workload cfg2 --processor Transformer --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_cfg2_transformer_bf16.json" 2>/dev/null
python3 bench.py --workload cfg2 --steps 20 --warmup 5 > "$OUT/bench_cfg2_bf16_cpu_baseline.json" 2>/dev/null
for f in $OUT/*.json; do python3 -c "
import json,sys
d=json.loads(open('$f').readline()); print('$f', d['ms_per_step'], d['roofline']['frac'], d.get('roofline_edge',{}).get('frac'), d.get('kernel_time_ms'))"; done
