#!/bin/bash
# evidence pass: kernel-trace stats of the bench command, the three PMC passes, the schedule breakdown
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_vgg_gpu.py tests/test_kernels_gpu.py tests/test_step_gpu.py -q -k "vgg or bn or batch_norm or step_fp32 or variants" > gpurun_out/c4_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/c4_pytest.log
timeout -k 10 200 python bench.py --steps 30 --warmup 4 --no-cpu-baseline > gpurun_out/c4_bench.json 2> gpurun_out/c4_bench.err; echo "bench rc=$?"; grep "timed region" gpurun_out/c4_bench.err
timeout -k 10 300 bash tools/prof_top.sh c4 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/c4_prof_top.log 2>&1; echo "prof rc=$?"
timeout -k 10 500 bash tools/pmc_collect.sh c4 > gpurun_out/c4_pmc.log 2>&1; echo "pmc rc=$?"; tail -16 gpurun_out/c4_pmc.log
