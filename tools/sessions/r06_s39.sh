#!/bin/bash
# round 6, session 39: every processor family x checkpointing x eager / one graph through the training step at bench sizes (does any route raise?)
set -u
out=gpurun_out/r06_s39; mkdir -p $out
{
for cfg in cfg2 cfg3; do
  for proc in GNN Transformer GraphTransformer; do
    for ck in 0 1; do
      for gr in 0 1; do
        [ "$cfg $proc" = "cfg3 GraphTransformer" ] && continue   # (r06_s37)
        echo "== $cfg $proc checkpoint=$ck graph=$gr"
        ANEMOI_AMD_CHECKPOINT=$ck TRAIN_BENCH_GRAPH=$gr timeout 600 python tools/train_step_bench.py $cfg 3 $proc 2>&1 | grep "forward + backward\|Error\|error" | tail -3
      done
    done
  done
done
echo "== Transformer cfg2, dropout 0.1, one graph"
TRAIN_BENCH_DROPOUT=0.1 TRAIN_BENCH_GRAPH=1 timeout 600 python tools/train_step_bench.py cfg2 3 Transformer 2>&1 | grep "forward + backward\|Error\|error" | tail -3
} > $out/train_matrix.txt 2>&1
cat $out/train_matrix.txt
