#!/bin/bash
# round 6, session 51: the attention forward + backward fuzzer, then the block fuzzers on four more seeds
set -u
out=gpurun_out/r06_s51; mkdir -p $out
for seed in 41 42; do
  FUZZ_ONLY=mhsa_backward timeout 1200 python tools/fuzz_ops.py 480 $seed > $out/fuzz_mhsa_bwd_seed$seed.txt 2>&1
  grep -v amdgpu $out/fuzz_mhsa_bwd_seed$seed.txt | cut -c1-260 | tail -n 14
done
for seed in 43 44 45 46; do
  FUZZ_ONLY=blocks timeout 1200 python tools/fuzz_ops.py 360 $seed > $out/fuzz_blocks_seed$seed.txt 2>&1
  grep -v amdgpu $out/fuzz_blocks_seed$seed.txt | cut -c1-260 | tail -n 14
done
