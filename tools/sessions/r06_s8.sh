#!/bin/bash
# round 6, session 8: the whole GPU suite twice, every failure kept (two different bf16-tolerance tests failed in two earlier whole-suite runs and pass alone)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_s8
mkdir -p "$OUT"
cd "$ROOT"
for i in 1 2; do
  SECONDS=0; timeout 1500 python3 -m pytest tests -m gpu -q --tb=short -rf > "$OUT/suite$i.txt" 2> "$OUT/suite$i.err"; echo "suite $i rc=$? ${SECONDS}s $(tail -n 1 "$OUT/suite$i.txt")"; grep "^FAILED\|^E  " "$OUT/suite$i.txt" | cut -c1-300 | head -20
done
