#!/bin/bash
# round 5, session 13: both slices of a destination in one wave (NS = 2) on the scheduled edge kernel
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s13
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "scheduled" > "$OUT/pytest_a.txt" 2>&1; tail -3 "$OUT/pytest_a.txt"
ANEMOI_AMD_EDGE_NS=2 timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "scheduled" > "$OUT/pytest_ns2.txt" 2>&1; tail -3 "$OUT/pytest_ns2.txt"
{
ANEMOI_AMD_EDGE_SCHED=0 timeout 300 python3 tools/edge_bench.py --set proc --iters 50 --save /tmp/e_proc.pt
for ns in 1 2; do for w in 3 4; do
echo "NS=$ns WGS=$w"; ANEMOI_AMD_EDGE_NS=$ns ANEMOI_AMD_EDGE_WGS=$w timeout 300 python3 tools/edge_bench.py --set proc --iters 50 --compare /tmp/e_proc.pt
done; done
echo "NS=2 U=2 WGS=3/4/5"
for w in 3 4 5; do ANEMOI_AMD_EDGE_NS=2 ANEMOI_AMD_EDGE_U=2 ANEMOI_AMD_EDGE_WGS=$w timeout 300 python3 tools/edge_bench.py --set proc --iters 50 --compare /tmp/e_proc.pt; done
echo "defaults"; timeout 300 python3 tools/edge_bench.py --set proc --iters 50 --compare /tmp/e_proc.pt
} > "$OUT/edge_ab.txt" 2>&1
grep -v amdgpu.ids "$OUT/edge_ab.txt"
