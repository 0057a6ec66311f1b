#!/bin/bash
# round 6, session 6: the LDS-tile edge kernel (bit identity, A/B against the scheduled kernel), whole-forward repeats under contention, the suite again
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_s6
mkdir -p "$OUT"
cd "$ROOT"
timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tiles_is_the_plain" > "$OUT/tiles_test.txt" 2>&1; echo "tiles test rc=$? $(tail -n 1 "$OUT/tiles_test.txt")"; grep -m5 "Error\|assert" "$OUT/tiles_test.txt" | cut -c1-200
for i in 1 2; do timeout 120 python3 tools/edge_bench.py --iters 50 --save "$OUT/sched.pt" 2>&1 | tail -n 1; done
for caps in "72 256" "104 320" "56 192" "88 288" "40 128"; do set -- $caps
  ANEMOI_AMD_EDGE_TILES=1 ANEMOI_AMD_EDGE_TILE_SRC=$1 ANEMOI_AMD_EDGE_TILE_EDGES=$2 timeout 120 python3 tools/edge_bench.py --iters 50 --compare "$OUT/sched.pt" 2>&1 | tail -n 2
done
ANEMOI_AMD_EDGE_TILES=1 timeout 120 python3 tools/edge_bench.py --iters 50 --graph o96_ico5 --channels 512 2>&1 | tail -n 2
timeout 120 python3 tools/edge_bench.py --iters 50 --graph o96_ico5 --channels 512 2>&1 | tail -n 1
(timeout 300 python3 tools/micro/forward_repeat.py cfg2 GraphTransformer 400 > "$OUT/rep_a.txt" 2>&1 &
 timeout 300 python3 tools/micro/forward_repeat.py cfg2 GraphTransformer 400 > "$OUT/rep_b.txt" 2>&1 &
 wait)
tail -n 1 "$OUT/rep_a.txt" "$OUT/rep_b.txt"
(timeout 300 python3 tools/micro/forward_repeat.py cfg2 GNN 300 > "$OUT/rep_c.txt" 2>&1 &
 timeout 300 python3 tools/micro/forward_repeat.py cfg3 GraphTransformer 60 > "$OUT/rep_d.txt" 2>&1 &
 wait)
tail -n 1 "$OUT/rep_c.txt" "$OUT/rep_d.txt"
SECONDS=0; timeout 1500 python3 -m pytest tests -m gpu -q --durations=15 > "$OUT/suite.txt" 2> "$OUT/suite.err"; echo "suite rc=$? ${SECONDS}s $(tail -n 1 "$OUT/suite.txt")"; grep "^FAILED\|encoder latent" "$OUT/suite.txt" | cut -c1-250
