#!/bin/bash
# round-end evidence: full GPU suite, default bench, kernel-trace stats, PMC passes, schedule breakdown, config-4 / inference benches
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/ev_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/ev_pytest.log
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/ev_bench.json 2> gpurun_out/ev_bench.err; echo "bench rc=$?"; cut -c1-200 gpurun_out/ev_bench.json
timeout -k 10 300 bash tools/prof_top.sh ev --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/ev_prof_top.log 2>&1; echo "prof rc=$?"
timeout -k 10 500 bash tools/pmc_collect.sh ev > gpurun_out/ev_pmc.log 2>&1; echo "pmc rc=$?"
timeout -k 10 200 python tools/step_breakdown.py > gpurun_out/ev_breakdown.log 2>&1; cat gpurun_out/ev_breakdown.log
timeout -k 10 200 python bench.py --config 4 --steps 10 --warmup 3 > gpurun_out/ev_bench_cfg4.json 2> /dev/null; cut -c1-200 gpurun_out/ev_bench_cfg4.json
timeout -k 10 200 python tools/bench_inference.py > gpurun_out/ev_inference_cfg5.json 2>/dev/null; cat gpurun_out/ev_inference_cfg5.json
