#!/bin/bash
# Lab builds of the kernel library with -D switches of csrc/gemm.hip (A/B runs: tools/micro/ab_gemm.sh picks up
# tools/micro/bin/libanemoi_amd_*.so).  Usage: build_gemm_labs.sh name "-DFOO=1 -DBAR=2" [name2 "..."] ...
set -e
cd "$(dirname "$0")/../.."
python3 -c "from anemoi_models_amd import _build; _build.build()"
OBJ=anemoi_models_amd/lib/obj
mkdir -p tools/micro/bin /tmp/gemm_labs
while [ $# -ge 2 ]; do
  name=$1; defs=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $defs \
    -c anemoi_models_amd/csrc/gemm.hip -o /tmp/gemm_labs/gemm_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/micro/bin/libanemoi_amd_$name.so \
    /tmp/gemm_labs/gemm_$name.o $(ls $OBJ/*.o | grep -v '/gemm.o$')
  echo "built tools/micro/bin/libanemoi_amd_$name.so ($defs)"
done
