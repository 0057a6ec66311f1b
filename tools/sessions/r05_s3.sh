#!/bin/bash
# round 5, session 3: scheduled edge kernel -- bit identity after the rounding fix; what the balance and the chain prefetch are
# worth separately; workgroups per CU
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s3
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "scheduled" > "$OUT/pytest_edge.txt" 2>&1; tail -5 "$OUT/pytest_edge.txt"
{
ANEMOI_AMD_EDGE_SCHED=0 timeout 300 python3 tools/edge_bench.py --set proc --iters 50 --save /tmp/e_proc.pt
ANEMOI_AMD_EDGE_SCHED=1 timeout 300 python3 tools/edge_bench.py --set proc --iters 50 --compare /tmp/e_proc.pt
echo "balance off:"
ANEMOI_AMD_EDGE_BALANCE=0 timeout 300 python3 tools/edge_bench.py --set proc --iters 50 --compare /tmp/e_proc.pt
echo "workgroups per CU 4 / 6 / 8 (U = 4: 88 VGPRs -> 5 resident; U = 3: 78 -> 6):"
for w in 4 6 8; do
ANEMOI_AMD_EDGE_WGS=$w timeout 300 python3 tools/edge_bench.py --set proc --iters 50 --compare /tmp/e_proc.pt
ANEMOI_AMD_EDGE_WGS=$w ANEMOI_AMD_EDGE_U=3 timeout 300 python3 tools/edge_bench.py --set proc --iters 50
done
echo "encoder, balance off:"
ANEMOI_AMD_EDGE_SCHED=0 timeout 300 python3 tools/edge_bench.py --set enc --iters 50 --save /tmp/e_enc.pt
ANEMOI_AMD_EDGE_BALANCE=0 timeout 300 python3 tools/edge_bench.py --set enc --iters 50 --compare /tmp/e_enc.pt
ANEMOI_AMD_EDGE_BALANCE=1 timeout 300 python3 tools/edge_bench.py --set enc --iters 50 --compare /tmp/e_enc.pt
} > "$OUT/edge_ab.txt" 2>&1
grep -v amdgpu.ids "$OUT/edge_ab.txt"
echo "PMC, scheduled kernel (mesh launch):" > "$OUT/edge_pmc_sched.txt"
ANEMOI_AMD_EDGE_SCHED=1 bash tools/edge_pmc.sh --set proc >> "$OUT/edge_pmc_sched.txt" 2>&1
echo "PMC, round-robin kernel (mesh launch):" > "$OUT/edge_pmc_plain.txt"
ANEMOI_AMD_EDGE_SCHED=0 bash tools/edge_pmc.sh --set proc >> "$OUT/edge_pmc_plain.txt" 2>&1
tail -40 "$OUT/edge_pmc_sched.txt"
