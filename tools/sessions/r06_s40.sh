#!/bin/bash
# round 6, session 40: the whole GPU suite as the driver runs it (-x), wall time and the slowest tests; then smoke() and the default bench line
set -u
out=gpurun_out/r06_s40; mkdir -p $out
SECONDS=0
timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=15 > $out/suite.txt 2>&1
echo "suite rc=$? wall ${SECONDS}s" >> $out/suite.txt
tail -n 30 $out/suite.txt
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; echo "smoke rc=$?" >> $out/smoke.txt; tail -n 3 $out/smoke.txt
python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"; cut -c1-600 $out/bench_default.json
