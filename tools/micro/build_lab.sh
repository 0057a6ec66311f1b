#!/bin/bash
# Lab build of the kernel library with -D switches of ONE source (A/B runs through tools/micro/run_with_lib.py):
#   build_lab.sh <source without .hip> <name> "-DFOO=1 -DBAR=2"   ->  tools/micro/bin/libanemoi_amd_<name>.so
set -e
cd "$(dirname "$0")/../.."
python3 -c "from anemoi_models_amd import _build; _build.build()"
OBJ=anemoi_models_amd/lib/obj
src=$1; name=$2; defs=$3
mkdir -p tools/micro/bin /tmp/anemoi_labs
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DANEMOI_HIPCC_VERSION='"lab"' $defs \
  -c anemoi_models_amd/csrc/$src.hip -o /tmp/anemoi_labs/${src}_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/micro/bin/libanemoi_amd_$name.so \
  /tmp/anemoi_labs/${src}_$name.o $(ls $OBJ/*.o | grep -v "/$src.o\$")
echo "built tools/micro/bin/libanemoi_amd_$name.so ($src: $defs)"
