#!/bin/bash
# round 6, session 38: bf16 Transformer-model training on assembled input rows (regression found by the training-step table), then the table's Transformer rows
set -u
out=gpurun_out/r06_s38; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_training.py -q -x -m gpu -k "transformer_model or head_size_4" > $out/test.txt 2>&1
echo "test rc=$?" >> $out/test.txt
tail -n 12 $out/test.txt
{
echo "== Transformer-processor model, config 3, no dropout / dropout 0.1"
ANEMOI_AMD_CHECKPOINT=0 python tools/train_step_bench.py cfg3 3 Transformer 2>&1 | grep "forward\|Error"
TRAIN_BENCH_DROPOUT=0.1 ANEMOI_AMD_CHECKPOINT=0 python tools/train_step_bench.py cfg3 3 Transformer 2>&1 | grep "forward\|Error"
} > $out/train_tfm.txt 2>&1
cat $out/train_tfm.txt
