#!/bin/bash
# round 6, session 36: GraphTransformerConv dropout (ABI v41) on the GPU + the two tests with on-failure diagnostics
set -u
out=gpurun_out/r06_s36; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "conv_module or conv_dropout or gt_conv or many_edge" > $out/conv.txt 2>&1
echo "conv rc=$?" >> $out/conv.txt
timeout 900 python -m pytest tests/test_gpu_training.py -q -x -m gpu -k "conv or explicit or folded_edge or without_edges or do_not_take" > $out/train.txt 2>&1
echo "train rc=$?" >> $out/train.txt
tail -n 15 $out/conv.txt $out/train.txt
