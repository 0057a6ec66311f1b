#!/bin/bash
# round 5, session 5: (a) the scheduled edge kernel after the fma epilogue: bit identity, workgroups per CU 2 / 3 / auto;
# (b) the 2-workgroups-per-CU build of the w4 attention pipeline (ANEMOI_AMD_MHSA_QB=2): correctness, ms, MFMA-busy counters
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s5
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "scheduled or folded or block_level" > "$OUT/pytest_edge.txt" 2>&1; tail -3 "$OUT/pytest_edge.txt"
{
for w in 2 3 0; do
ANEMOI_AMD_EDGE_WGS=$w timeout 300 python3 tools/edge_bench.py --set proc --iters 50
done
ANEMOI_AMD_EDGE_SCHED=0 timeout 300 python3 tools/edge_bench.py --set proc --iters 50
ANEMOI_AMD_EDGE_WGS=0 timeout 300 python3 tools/edge_bench.py --graph o96_ico5 --channels 512 --set proc --iters 50
ANEMOI_AMD_EDGE_SCHED=0 timeout 300 python3 tools/edge_bench.py --graph o96_ico5 --channels 512 --set proc --iters 50
} > "$OUT/edge_ab.txt" 2>&1
grep -v amdgpu.ids "$OUT/edge_ab.txt"
{
echo "== QB = 4 (shipped)"; ANEMOI_AMD_MHSA_QB=4 timeout 600 python3 tools/mhsa_bench.py
echo "== QB = 2 (two workgroups per CU)"; ANEMOI_AMD_MHSA_QB=2 timeout 600 python3 tools/mhsa_bench.py
echo "== QB = 2, S = 8192"; ANEMOI_AMD_MHSA_QB=2 timeout 600 python3 tools/mhsa_bench.py 8192
echo "== QB = 2, S = 1000"; ANEMOI_AMD_MHSA_QB=2 timeout 600 python3 tools/mhsa_bench.py 1000
} > "$OUT/mhsa_qb.txt" 2>&1
grep -v amdgpu.ids "$OUT/mhsa_qb.txt"
ANEMOI_AMD_MHSA_QB=2 timeout 900 python3 -m pytest tests/test_gpu_attention_sizes.py -x -q -m gpu > "$OUT/pytest_attn_qb2.txt" 2>&1; tail -3 "$OUT/pytest_attn_qb2.txt"
ANEMOI_AMD_MHSA_QB=2 timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mhsa or attention or transformer" > "$OUT/pytest_attn_qb2_parity.txt" 2>&1; tail -3 "$OUT/pytest_attn_qb2_parity.txt"
