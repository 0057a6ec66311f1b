#!/bin/bash
# round 6, session 2: the moved bf16 anchor (latent 6.5e-3 -> 1.2e-2?), the seam cost a chained GEMM could recover, ABI v40 checks
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_s2
mkdir -p "$OUT"
cd "$ROOT"
timeout 600 python3 tools/micro/latent_anchor_diag.py 16 128 > "$OUT/latent_diag.txt" 2>&1; echo "diag rc=$?"; grep threads "$OUT/latent_diag.txt" | cut -c1-400
timeout 300 python3 tools/micro/chain_upper_bound.py > "$OUT/chain_upper_bound.txt" 2>&1; echo "chain rc=$?"; cat "$OUT/chain_upper_bound.txt"
timeout 900 python3 -m pytest tests/test_abi.py tests/test_bench_contract.py tests/test_gpu_parity.py -m gpu -x -q -k "abi or secondary or sched or block_level or edge_attention" > "$OUT/abi40.txt" 2>&1; echo "abi40 rc=$? $(tail -n 1 "$OUT/abi40.txt")"
timeout 300 python3 tools/sim_rank.py --worlds 8 --ranks 0,3,7 --detail > "$OUT/sim_rank8.txt" 2>&1; echo "sim rc=$?"; grep -v "^linear\|^gt_edge" "$OUT/sim_rank8.txt" | tail -n 6
