#!/bin/bash
# round 6, session 41: the hierarchical model at O96 / three levels against the oracle (f32, bf16) and a bf16 training step
set -u
out=gpurun_out/r06_s41; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_baseline_sizes.py -q -x -s -m gpu -k "hierarchical or all_gnn" --durations=3 > $out/hier.txt 2>&1
echo "rc=$?" >> $out/hier.txt
tail -n 40 $out/hier.txt | cut -c1-300
