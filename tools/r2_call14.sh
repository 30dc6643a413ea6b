#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
D=$PWD/pytorch-tecogan_amd/csrc
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py tests/test_conv3_rw_gpu.py -q -x > gpurun_out/c14_pytest.log 2>&1; tail -2 gpurun_out/c14_pytest.log
for lib in "$D/libtecogan_hip_noslim.so" "" "$D/libtecogan_hip_noslim.so" ""; do
  echo "== lib=$lib"
  TECOGAN_LIB=$lib timeout -k 10 200 python tools/step_breakdown.py 2>&1 | grep -E "alone|whole step" | grep -v "prep\|update\|lane"
done
