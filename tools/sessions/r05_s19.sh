#!/bin/bash
# round 5, session 19: ordered kernel trace of one config-3 GraphTransformer training step
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_s19
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export ANEMOI_AMD_CHECKPOINT=0
rm -rf /tmp/ktb
rocprofv3 --kernel-trace --output-format csv -d /tmp/ktb -o kt -- python3 $ROOT/tools/train_step_bench.py cfg3 1 > "$OUT/step_prof.log" 2>&1
python3 - <<'P' > "$OUT/ordered.txt"
import csv, glob
f = glob.glob('/tmp/ktb/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
for r in rows:
    s = int(r['Start_Timestamp']); e = int(r['End_Timestamp'])
    print(f"{(s-t0)/1e3:12.1f} {(e-s)/1e3:9.1f} {int(r['Grid_Size_X'])//max(1,int(r['Workgroup_Size_X'])):7d} {r['Kernel_Name'][:150]}")
P
wc -l "$OUT/ordered.txt"
