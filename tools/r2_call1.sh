#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/c1_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/c1_pytest.log
tail -3 gpurun_out/c1_pytest.log
timeout -k 10 200 python tools/mb_rw.py > gpurun_out/c1_mb_rw.log 2>&1; echo "mb_rw rc=$?"
timeout -k 10 120 python bench.py --steps 30 --warmup 4 --no-cpu-baseline > gpurun_out/c1_bench.json 2> gpurun_out/c1_bench.err; echo "bench rc=$?"
TECOGAN_RW=1 timeout -k 10 120 python bench.py --steps 30 --warmup 4 --no-cpu-baseline > gpurun_out/c1_bench_rw.json 2> gpurun_out/c1_bench_rw.err; echo "bench rw rc=$?"
timeout -k 10 200 bash tools/prof_top.sh c1 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/c1_prof_top.log 2>&1; echo "prof rc=$?"
cut -c1-300 gpurun_out/c1_bench.json; echo; cut -c1-300 gpurun_out/c1_bench_rw.json
