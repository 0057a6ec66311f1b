#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_s11
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_attention_sizes.py tests/test_gpu_training.py -x -q -m gpu -k "mhsa or transformer or attention" > "$OUT/pytest_mhsa.txt" 2>&1
tail -3 "$OUT/pytest_mhsa.txt"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ktt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktt -o kt -- python3 "$ROOT/bench.py" --processor Transformer --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench_tfm_prof.txt" 2>&1
python3 "$ROOT/tools/summarize_trace.py" /tmp/ktt | grep -E "mhsa|transpose_v" | cut -c1-40,100-170
cd "$ROOT"
python3 bench.py --processor Transformer --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_tfm.txt" 2>&1
python3 - <<'PY'
import json
d=json.loads([x for x in open("gpurun_out/r04_s11/bench_tfm.txt") if x.startswith("{")][-1]); print(d["ms_per_step"], d["roofline_mhsa"]["frac"], d["roofline_mhsa"]["avg_launch_ms"])
PY
