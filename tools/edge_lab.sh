#!/bin/bash
# A/B of the edge kernel's pipelining levels on the three edge sets of config 3 (run on the GPU box):
#   gpurun --timeout 900 -- 'bash tools/edge_lab.sh > gpurun_out/edge_lab.txt 2>&1'
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
for set in proc dec enc; do
  ANEMOI_AMD_EDGE_PIPE=0 python3 tools/edge_bench.py --set $set --iters 20 --save /tmp/edge_$set.pt 2>&1 | grep -v amdgpu.ids
  for pipe in 1 2 3 12 13; do
    ANEMOI_AMD_EDGE_PIPE=$pipe python3 tools/edge_bench.py --set $set --iters 20 --compare /tmp/edge_$set.pt 2>&1 | grep -v amdgpu.ids
  done
done
for wgs in 3 4 6 8; do
  ANEMOI_AMD_EDGE_WGS=$wgs ANEMOI_AMD_EDGE_PIPE=3 python3 tools/edge_bench.py --set proc --iters 20 2>&1 | grep -v amdgpu.ids | sed "s/^/WGS=$wgs /"
  ANEMOI_AMD_EDGE_WGS=$wgs ANEMOI_AMD_EDGE_PIPE=3 python3 tools/edge_bench.py --set dec --iters 20 2>&1 | grep -v amdgpu.ids | sed "s/^/WGS=$wgs /"
done
ANEMOI_AMD_EDGE_PIPE=3 python3 tools/edge_bench.py --set proc --col near --iters 20 2>&1 | grep -v amdgpu.ids
ANEMOI_AMD_EDGE_PIPE=0 python3 tools/edge_bench.py --set proc --col near --iters 20 2>&1 | grep -v amdgpu.ids
