#!/bin/bash
# round 6, session 49: the fuzzers on four more seeds, more cases
set -u
out=gpurun_out/r06_s49; mkdir -p $out
for seed in 21 22 23 24; do
  timeout 1500 python tools/fuzz_ops.py 480 $seed > $out/fuzz_seed$seed.txt 2>&1
  echo "seed $seed rc=$?" >> $out/fuzz_seed$seed.txt
  grep -v amdgpu $out/fuzz_seed$seed.txt | grep "bad of\|  " | cut -c1-240 | tail -n 24
done
