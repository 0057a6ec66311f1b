#!/bin/bash
# round 6, session 43: configs 2 and 5 eager against one HIP graph per forward (is the small-mesh step bound by the host's enqueue?)
set -u
out=gpurun_out/r06_s43; mkdir -p $out
for wl in "cfg2 GraphTransformer" "cfg2 GNN" "cfg2 Transformer" "cfg1 GraphTransformer"; do
  set -- $wl
  for mode in "" "--hipgraph"; do
    python bench.py --workload $1 --processor $2 --no-cpu-baseline --no-secondary --steps 50 --warmup 10 $mode 2> /dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1 $2 ${mode:-eager}', d['ms_per_step'], d.get('host_enqueue_ms'), d['roofline']['frac'])"
  done
done 2>&1 | tee $out/graph_vs_eager.txt
