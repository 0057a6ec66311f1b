#!/bin/bash
# round 6, session 44: per-kernel event times of the bench line, instrumented forward enqueued as before (hold 0) / by the shipped rule (8-ms hold for steps under 10 ms)
set -u
out=gpurun_out/r06_s44; mkdir -p $out
for hold in 0 rule; do
  for wl in "cfg2 GraphTransformer" "cfg2 GNN" "cfg1 GraphTransformer" "cfg3 GraphTransformer"; do
    set -- $wl
    if [ $hold = rule ]; then unset ANEMOI_AMD_BENCH_HOLD_MS; else export ANEMOI_AMD_BENCH_HOLD_MS=$hold; fi
    python bench.py --workload $1 --processor $2 --no-cpu-baseline --no-secondary 2> /dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('hold $hold: $1 $2', d['ms_per_step'], 'linear frac', d['roofline']['frac'], 'avg launch', d['roofline']['avg_launch_ms'], 'edge', d.get('roofline_edge',{}).get('frac'), d.get('kernel_time_ms'))"
  done
done 2>&1 | tee $out/hold_rule.txt
