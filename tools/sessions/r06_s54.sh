#!/bin/bash
# round 6, session 54: every fuzzer on three more seeds (final library)
set -u
out=gpurun_out/r06_s54; mkdir -p $out
for seed in 61 62 63; do
  timeout 1500 python tools/fuzz_ops.py 240 $seed > $out/fuzz_seed$seed.txt 2>&1
  echo "seed $seed rc=$?" >> $out/fuzz_seed$seed.txt
  grep -v amdgpu $out/fuzz_seed$seed.txt | grep "bad of\|^  " | cut -c1-260 | tail -n 20
done
