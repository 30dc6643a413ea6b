#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
D=$PWD/pytorch-tecogan_amd/csrc
for lib in "" "$D/libtecogan_hip_s1.so" "$D/libtecogan_hip_s2.so" ""; do
  echo "== lib=$lib"
  TECOGAN_LIB=$lib timeout -k 10 200 python tools/step_breakdown.py 2>&1 | grep -E "g_bwd alone|whole step"
  TECOGAN_LIB=$lib timeout -k 10 200 python tools/microbench.py wgrad 2>&1 | cut -c1-84 | grep -v amdgpu
done
