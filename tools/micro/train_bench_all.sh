# the training-step table of profiles/rNN_train_step_bench.txt: bash tools/micro/train_bench_all.sh   (through gpurun)
for cfg in cfg2 cfg3; do
  for ck in 0 1; do
    echo "== $cfg checkpoint=$ck"
    ANEMOI_AMD_CHECKPOINT=$ck python tools/train_step_bench.py $cfg 8 2>&1 | grep "forward"
    ANEMOI_AMD_CHECKPOINT=$ck TRAIN_BENCH_GRAPH=1 python tools/train_step_bench.py $cfg 8 2>&1 | grep "forward"
  done
done
echo "== inference forward of the same box"
python bench.py --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
