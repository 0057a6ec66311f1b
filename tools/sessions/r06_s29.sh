#!/bin/bash
# round 6, session 29: the new bit-identity test of the staggered GEMM start
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "staggered_start" 2>&1 | tail -n 5
