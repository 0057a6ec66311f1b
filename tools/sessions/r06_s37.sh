#!/bin/bash
# round 6, session 37: the training-step table on round 6's library (profiles/r06_train_step_bench.txt), with the GEMM stagger off / on
set -u
out=gpurun_out/r06_s37; mkdir -p $out
{
bash tools/micro/train_bench_all.sh
echo "== cfg3, un-checkpointed, one HIP graph, GEMM stagger off / default, alternated"
for i in 1 2; do
  for st in "0,0,2" "2,16,2"; do
    echo "stagger $st: $(ANEMOI_AMD_GEMM_STAGGER=$st ANEMOI_AMD_CHECKPOINT=0 TRAIN_BENCH_GRAPH=1 python tools/train_step_bench.py cfg3 8 2>&1 | grep 'HIP graph')"
  done
done
echo "== Transformer-processor model, config 3, no dropout / dropout 0.1"
ANEMOI_AMD_CHECKPOINT=0 python tools/train_step_bench.py cfg3 3 Transformer 2>&1 | grep "forward"
TRAIN_BENCH_DROPOUT=0.1 ANEMOI_AMD_CHECKPOINT=0 python tools/train_step_bench.py cfg3 3 Transformer 2>&1 | grep "forward"
echo "== attention alone (tools/mhsa_bwd_bench.py)"
python tools/mhsa_bwd_bench.py 2>&1 | tail -4
} > $out/train_step_bench.txt 2>&1
cat $out/train_step_bench.txt
