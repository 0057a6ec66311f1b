#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_s9
mkdir -p "$OUT"
cd "$ROOT"
timeout 3000 python3 -m pytest tests -x -q -m gpu > "$OUT/pytest_gpu.txt" 2>&1
tail -4 "$OUT/pytest_gpu.txt"
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --detail > "$OUT/bench_cfg3.txt" 2>&1
grep -o '"ms_per_step": [0-9.]*' "$OUT/bench_cfg3.txt"; grep "edges=" "$OUT/bench_cfg3.txt" | cut -c1-110
python3 - <<'PY'
import json
l=[x for x in open("gpurun_out/r04_s9/bench_cfg3.txt") if x.startswith("{")][-1]
d=json.loads(l); print(d["roofline_edge"]["frac"], d["roofline_edge"]["avg_launch_ms"], d["roofline"]["frac"])
PY
python3 bench.py --rollout 4 --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/bench_cfg4.txt" 2>&1
grep -o '"ms_per_step": [0-9.]*' "$OUT/bench_cfg4.txt"
python3 tools/sim_rank.py --worlds 8 --steps 10 > "$OUT/sim_rank.txt" 2>&1; tail -1 "$OUT/sim_rank.txt"
