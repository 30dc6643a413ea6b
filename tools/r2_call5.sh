#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py tests/test_bench_config_gpu.py -q -x > gpurun_out/c5_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/c5_pytest.log
timeout -k 10 200 python bench.py --steps 40 --warmup 4 --no-cpu-baseline > gpurun_out/c5_bench.json 2> gpurun_out/c5_bench.err; echo "bench rc=$?"; grep "timed region" gpurun_out/c5_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c5_bench.json').read().strip().splitlines()[-1])
for k,v in d['roofline']['hbm_kernels'].items(): print(f"{k:40s} {v}")
PY
