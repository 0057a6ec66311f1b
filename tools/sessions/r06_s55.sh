#!/bin/bash
# round 6, session 55: bench.py over its flag combinations at the small workloads (does any combination raise or print a malformed line?)
set -u
out=gpurun_out/r06_s55; mkdir -p $out
: > $out/bench_matrix.txt
for wl in cfg1 cfg2; do
  for proc in GraphTransformer GNN Transformer; do
    for dt in bf16 fp32; do
      for extra in "" "--hipgraph" "--rollout 3" "--rollout 2 --hipgraph"; do
        line=$(timeout 600 python bench.py --workload $wl --processor $proc --dtype $dt --steps 3 --warmup 1 --no-cpu-baseline --no-secondary $extra 2> $out/err.txt | tail -n 1)
        rc=$?
        echo "$wl $proc $dt [$extra] -> $(echo "$line" | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.readline())
    print('ms', d['ms_per_step'], 'roofline', d.get('roofline', {}).get('frac'), 'identical', d.get('run_to_run_identical'))
except Exception as exc:
    print('NO JSON LINE:', type(exc).__name__)
") $(tail -n 1 $out/err.txt | cut -c1-160 | grep -i "error" )" >> $out/bench_matrix.txt
      done
    done
  done
done
cat $out/bench_matrix.txt
