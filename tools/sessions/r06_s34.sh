#!/bin/bash
# round 6, session 34: the Transformer-processor model at config 2's size against the oracle (new parity case)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
SECONDS=0; timeout 1500 python3 -m pytest tests/test_gpu_baseline_sizes.py -m gpu -q -x -s -k "transformer_processor_16_blocks" 2>&1 | grep -v amdgpu | tail -n 8; echo "${SECONDS}s"
