#!/bin/bash
# round 5, session 32: bench.py at N = 2 and 4 exactly as the driver launches it (torch.distributed.run), the ranks sharing this one GPU
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s32
mkdir -p "$OUT"
cd "$ROOT"
for n in 2 4; do
  ANEMOI_AMD_BENCH_SHARE_GPU=1 timeout 700 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600 + n)) bench.py --gpus $n --steps 3 --warmup 1 > "$OUT/bench_world$n.txt" 2> "$OUT/bench_world$n.err"
  echo "N=$n rc=$?"; grep -a '^{' "$OUT/bench_world$n.txt" | cut -c1-330
done
