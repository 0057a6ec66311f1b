#!/bin/bash
# round 6, session 46: differential fuzzing on the final library, two new seeds (adds the conv with dropout, forward + backward, and the attention forward)
set -u
out=gpurun_out/r06_s46; mkdir -p $out
for seed in 11 12; do
  timeout 1200 python tools/fuzz_ops.py 240 $seed > $out/fuzz_seed$seed.txt 2>&1
  echo "seed $seed rc=$?" >> $out/fuzz_seed$seed.txt
  grep -v amdgpu $out/fuzz_seed$seed.txt | tail -n 14
done
