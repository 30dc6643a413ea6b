set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_fp16_gpu.py tests/test_dp_gpu.py tests/test_step_gpu.py tests/test_module_state_gpu.py -q -x > gpurun_out/c1_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/c1_pytest.log
timeout -k 10 200 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-300
