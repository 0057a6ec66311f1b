#!/bin/bash
# A/B of the GEMM epilogue variants on the GPU box (lab; not part of the product):
#   gpurun -- 'bash tools/micro/ab_gemm.sh'
# product library against lab builds of the same sources in tools/micro/bin/ (see the -D switches at the top of csrc/gemm.hip)
OUT=gpurun_out/ab_gemm
mkdir -p $OUT
SHAPES=${SHAPES:-"40962x4096x1024 40962x4288x1024 40962x1024x4096 40962x1024x1216 542080x4096x1024 542080x2048x256 5121x4096x1024 40960x4096x8192"}
python3 tools/gemm_check.py > $OUT/check_product.txt 2>&1
tail -1 $OUT/check_product.txt
LIBS="product $(ls tools/micro/bin/libanemoi_amd_*.so 2>/dev/null)"
for rep in 1 2; do
  for lib in $LIBS; do
    name=$(basename $lib .so)
    if [ "$lib" = product ]; then arg=""; else arg="--lib $lib"; fi
    GEMM_BENCH_BLASLT=0 python3 tools/gemm_bench.py $arg $SHAPES > $OUT/bench_${name}_$rep.txt 2>&1
  done
done
for lib in $LIBS; do
  name=$(basename $lib .so)
  echo "== $name (TFLOP/s, two passes)"
  paste <(grep act= $OUT/bench_${name}_1.txt | awk '{print $2,$4,$6,$7,$8,$11}') <(grep act= $OUT/bench_${name}_2.txt | awk '{print $11}')
done
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --detail > $OUT/bench_step.txt 2>&1
tail -1 $OUT/bench_step.txt | cut -c1-300
