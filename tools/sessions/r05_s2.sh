#!/bin/bash
# round 5, session 2: the scheduled edge kernel -- bit identity, A/B against the round-robin kernel on the three edge sets
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s2
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "scheduled or folded" > "$OUT/pytest_edge.txt" 2>&1; tail -5 "$OUT/pytest_edge.txt"
{
for set in proc enc; do
  ANEMOI_AMD_EDGE_SCHED=0 timeout 300 python3 tools/edge_bench.py --set $set --iters 50 --save /tmp/e_$set.pt
  for u in 4 6 3; do
    ANEMOI_AMD_EDGE_SCHED=1 ANEMOI_AMD_EDGE_U=$u timeout 300 python3 tools/edge_bench.py --set $set --iters 50 --compare /tmp/e_$set.pt
  done
done
ANEMOI_AMD_EDGE_RUNS=1 timeout 300 python3 tools/edge_bench.py --set dec --iters 30 --save /tmp/e_dec.pt
for u in 4 3; do
ANEMOI_AMD_EDGE_RUNS=0 ANEMOI_AMD_EDGE_SCHED=1 ANEMOI_AMD_EDGE_U=$u timeout 300 python3 tools/edge_bench.py --set dec --iters 30 --compare /tmp/e_dec.pt
done
ANEMOI_AMD_EDGE_RUNS=0 ANEMOI_AMD_EDGE_SCHED=0 timeout 300 python3 tools/edge_bench.py --set dec --iters 30 --compare /tmp/e_dec.pt
# the O96 / ico-5 sets (config 2)
for set in proc enc; do
  ANEMOI_AMD_EDGE_SCHED=0 timeout 300 python3 tools/edge_bench.py --graph o96_ico5 --channels 512 --set $set --iters 50 --save /tmp/e2_$set.pt
  ANEMOI_AMD_EDGE_SCHED=1 timeout 300 python3 tools/edge_bench.py --graph o96_ico5 --channels 512 --set $set --iters 50 --compare /tmp/e2_$set.pt
done
} > "$OUT/edge_ab.txt" 2>&1
grep -v amdgpu.ids "$OUT/edge_ab.txt"
