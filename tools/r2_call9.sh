#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
A1=$PWD/pytorch-tecogan_amd/csrc/libtecogan_hip_a1.so
for lib in "$A1" "" "$A1" ""; do
  echo "== lib=$lib"
  TECOGAN_LIB=$lib timeout -k 10 200 python tools/step_breakdown.py 2>&1 | grep -E "g_bwd alone|d_real alone|d_fake_bwd alone|whole step"
done
