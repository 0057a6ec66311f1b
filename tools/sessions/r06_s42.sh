#!/bin/bash
# round 6, session 42: the whole GPU suite three times on one box (does any run fail a test that passes in the others?)
set -u
out=gpurun_out/r06_s42; mkdir -p $out
for i in 1 2 3; do
  SECONDS=0
  timeout 1200 python -m pytest tests/ -q -m gpu > $out/suite_$i.txt 2>&1
  echo "suite $i rc=$? wall ${SECONDS}s" >> $out/suite_$i.txt
  tail -n 3 $out/suite_$i.txt
  grep -n "FAILED\|Error" $out/suite_$i.txt | head -5
done
