#!/bin/bash
# round 6, session 35: whole-forward bit identity when the kernels start cold (idle GPU, busy host, evicted caches, f32 route in between)
set -u
out=gpurun_out/r06_s35; mkdir -p $out
timeout 900 python tools/micro/cold_forward_repeat.py cfg2 GraphTransformer 60 > $out/cold_cfg2.txt 2>&1
echo "cfg2 rc=$?" >> $out/cold_cfg2.txt
timeout 900 python tools/micro/cold_forward_repeat.py cfg3 GraphTransformer 25 > $out/cold_cfg3.txt 2>&1
echo "cfg3 rc=$?" >> $out/cold_cfg3.txt
tail -n 30 $out/cold_cfg2.txt $out/cold_cfg3.txt
