#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
D=$PWD/pytorch-tecogan_amd/csrc
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -q -k wgrad 2>&1 | tail -2
TECOGAN_LIB=$D/libtecogan_hip_stamp.so timeout -k 10 100 python tools/stamp_wgrad.py 2>&1 | grep -v amdgpu
for lib in "$D/libtecogan_hip_orig.so" "" "$D/libtecogan_hip_orig.so" ""; do
  echo "== lib=$lib"
  TECOGAN_LIB=$lib timeout -k 10 200 python tools/step_breakdown.py 2>&1 | grep -E "g_bwd alone|d_real alone|whole step"
  TECOGAN_LIB=$lib timeout -k 10 200 python tools/microbench.py wgrad 2>&1 | cut -c1-84 | grep -v amdgpu
done
