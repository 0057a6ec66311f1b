#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s14
mkdir -p "$OUT"
cd "$ROOT"
{
for col in graph near random; do
echo "col=$col"
ANEMOI_AMD_EDGE_SCHED=0 timeout 300 python3 tools/edge_bench.py --set proc --iters 50 --col $col
ANEMOI_AMD_EDGE_NS=1 timeout 300 python3 tools/edge_bench.py --set proc --iters 50 --col $col
ANEMOI_AMD_EDGE_NS=2 timeout 300 python3 tools/edge_bench.py --set proc --iters 50 --col $col
done
echo "natural (level-by-level) mesh order instead of Morton:"
ANEMOI_AMD_EDGE_NS=1 timeout 300 python3 tools/edge_bench.py --set proc --iters 50 --order natural
} > "$OUT/edge_col.txt" 2>&1
grep -v amdgpu.ids "$OUT/edge_col.txt"
cd /tmp && export TMPDIR=/tmp
for ns in 1 2; do
for ctr in "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf /tmp/pm
  ANEMOI_AMD_EDGE_NS=$ns rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pm -- python3 $ROOT/tools/edge_bench.py --iters 5 --set proc > /tmp/pm.log 2>&1
  echo "NS=$ns" >> "$OUT/edge_pmc_ns.txt"; python3 $ROOT/tools/pmc_summary.py /tmp/pm gt_edge | sed -E 's/^.*(folded|sched)_kernel[^ ]* *[a-z, 0-9]*, /  /' >> "$OUT/edge_pmc_ns.txt"
done; done
cat "$OUT/edge_pmc_ns.txt"
