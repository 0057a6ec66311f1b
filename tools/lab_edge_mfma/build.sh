#!/bin/bash
# Builds the lab kernel into tools/lab_edge_mfma/_build/liblab_edge_mfma.so (git-ignored; travels with gpurun).
#   bash tools/lab_edge_mfma/build.sh [-DET_PROF=1]
set -e
here="$(cd "$(dirname "$0")" && pwd)"
mkdir -p "$here/_build"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 "$@" "$here/edge_attention_mfma.hip" -o "$here/_build/liblab_edge_mfma.so"
echo "built $here/_build/liblab_edge_mfma.so"
