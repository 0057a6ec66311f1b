#!/bin/bash
# round 6, session 45: bench.py --gpus N at config 3 (the driver's default workload) with the ranks sharing this GPU: protocol transcript, parity_vs_single -- not a measurement
set -u
out=gpurun_out/r06_s45; mkdir -p $out
for world in 2 8; do
  PORT=$((29500 + RANDOM % 400))
  SECONDS=0
  for r in $(seq 0 $((world - 1))); do
    ANEMOI_AMD_BENCH_SHARE_GPU=1 WORLD_SIZE=$world RANK=$r LOCAL_RANK=$r MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT \
      timeout 1200 python3 bench.py --gpus $world --steps 2 --warmup 1 --no-cpu-baseline > $out/world${world}.rank$r.txt 2>&1 &
  done
  wait
  echo "world $world: ${SECONDS}s"
  grep -o '"ms_per_step": [0-9.]*\|"parity_vs_single": {[^}]*}\|"n_gpus": [0-9]*' $out/world${world}.rank0.txt | head -5
  tail -n 2 $out/world${world}.rank1.txt | cut -c1-300
done
