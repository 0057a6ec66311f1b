#!/bin/bash
# round 6, session 50: the model-level fuzzer (whole forward / backward against the oracle over random configurations)
set -u
out=gpurun_out/r06_s50; mkdir -p $out
for seed in 31 32; do
  FUZZ_ONLY=models timeout 1500 python tools/fuzz_ops.py 480 $seed > $out/fuzz_models_seed$seed.txt 2>&1
  echo "seed $seed rc=$?" >> $out/fuzz_models_seed$seed.txt
  grep -v amdgpu $out/fuzz_models_seed$seed.txt | cut -c1-300 | tail -n 30
done
