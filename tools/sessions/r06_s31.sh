#!/bin/bash
# round 6, session 31: rocprof kernel summaries of the secondary workloads (config 2, config 5); cycle counters of the GEMM launches with the stagger off / on
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_s31
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for w in "cfg2 GraphTransformer" "cfg2 GNN"; do set -- $w
  rm -rf /tmp/kt2
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -o kt -- python3 "$ROOT/bench.py" --workload $1 --processor $2 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > "$OUT/rocprof_$1_$2.log" 2>&1
  python3 "$ROOT/tools/summarize_trace.py" /tmp/kt2 > "$OUT/kernel_summary_$1_$2.txt" 2>&1
  head -n 8 "$OUT/kernel_summary_$1_$2.txt" | cut -c1-150
done
for cfg in 0,0,2 2,16,2 0,0,2 2,16,2; do
  rm -rf /tmp/pmg
  ANEMOI_AMD_GEMM_STAGGER=$cfg rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY -d /tmp/pmg -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$OUT/pmc_$cfg.log" 2>&1
  echo "== stagger $cfg: $(grep -o '"ms_per_step": [0-9.]*' "$OUT/pmc_$cfg.log" | head -1)" | tee -a "$OUT/pmc_stagger.txt"
  python3 "$ROOT/tools/pmc_summary.py" /tmp/pmg linear_bf16_w4 | awk '{k=$(NF-3); n=$(NF-1); sub("n=","",n); v=$NF; sub("mean=","",v); s[k]+=n*v} END {for (k in s) printf "   %s total over the run %.4g\n", k, s[k]}' | tee -a "$OUT/pmc_stagger.txt"
done
