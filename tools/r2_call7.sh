#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
A1=$PWD/pytorch-tecogan_amd/csrc/libtecogan_hip_a1.so
echo "== wgrad microbench, AHEAD=1" > gpurun_out/c7_wgrad.log
TECOGAN_LIB=$A1 timeout -k 10 200 python tools/microbench.py wgrad >> gpurun_out/c7_wgrad.log 2>&1
echo "== wgrad microbench, AHEAD=2" >> gpurun_out/c7_wgrad.log
timeout -k 10 200 python tools/microbench.py wgrad >> gpurun_out/c7_wgrad.log 2>&1
cat gpurun_out/c7_wgrad.log
for lib in "$A1" ""; do
  TECOGAN_LIB=$lib timeout -k 10 120 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib=$lib', d['ms_per_step'], d['final_losses'])"
done
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -q -k wgrad 2>&1 | tail -2
