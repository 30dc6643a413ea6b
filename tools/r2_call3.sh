#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_module_state_gpu.py tests/test_vgg_gpu.py tests/test_fp16_gpu.py tests/test_inference_gpu.py -q > gpurun_out/c3_pytest.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/c3_pytest.log
timeout -k 10 200 python bench.py --steps 30 --warmup 4 > gpurun_out/c3_bench.json 2> gpurun_out/c3_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/c3_bench.err
timeout -k 10 200 python bench.py --config 4 --steps 10 --warmup 3 > gpurun_out/c3_bench_cfg4.json 2> gpurun_out/c3_bench_cfg4.err; echo "bench cfg4 rc=$?"; tail -3 gpurun_out/c3_bench_cfg4.err; cut -c1-420 gpurun_out/c3_bench_cfg4.json
timeout -k 10 200 python bench.py --config 4 --dtype bf16 --steps 10 --warmup 3 --no-roofline > gpurun_out/c3_bench_cfg4_bf16.json 2> gpurun_out/c3_bench_cfg4_bf16.err; echo "bench cfg4 bf16 rc=$?"; cut -c1-300 gpurun_out/c3_bench_cfg4_bf16.json
timeout -k 10 200 python tools/bench_inference.py > gpurun_out/c3_inference_cfg5.json 2>/dev/null; cat gpurun_out/c3_inference_cfg5.json
timeout -k 10 200 python tools/bench_inference.py --lr 32 --frames 10 > gpurun_out/c3_inference_cfg1.json 2>/dev/null; cat gpurun_out/c3_inference_cfg1.json
