#!/bin/bash
# round 6, session 53: final validation -- the whole GPU suite twice as the driver runs it (-x), smoke(), the default bench line
set -u
out=gpurun_out/r06_s53; mkdir -p $out
for i in 1 2; do
  SECONDS=0
  timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=12 > $out/suite_$i.txt 2>&1
  echo "suite $i rc=$? wall ${SECONDS}s" >> $out/suite_$i.txt
  tail -n 4 $out/suite_$i.txt | cut -c1-200
done
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; echo "smoke rc=$?" >> $out/smoke.txt; tail -n 2 $out/smoke.txt
python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"; cut -c1-400 $out/bench_default.json
