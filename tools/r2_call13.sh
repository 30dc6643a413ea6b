#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
D=$PWD/pytorch-tecogan_amd/csrc
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -q -k wgrad > gpurun_out/c13_pytest.log 2>&1; tail -2 gpurun_out/c13_pytest.log
for lib in "$D/libtecogan_hip_prev.so" "" "$D/libtecogan_hip_prev.so" ""; do
  echo "== lib=$lib"
  TECOGAN_LIB=$lib timeout -k 10 200 python tools/step_breakdown.py 2>&1 | grep -E "g_bwd alone|whole step"
  TECOGAN_LIB=$lib timeout -k 10 200 python tools/microbench.py wgrad 2>&1 | cut -c1-84 | grep -v amdgpu | head -5
done
