#!/bin/bash
# what the driver runs at round end: the GPU suite, smoke(), the default bench
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/full_pytest.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/full_pytest.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/full_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/full_smoke.log
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/full_bench.json 2> gpurun_out/full_bench.err; echo "bench rc=$?"; cut -c1-260 gpurun_out/full_bench.json
