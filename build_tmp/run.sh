cd "${GRAFT_REPO_ROOT:?}"
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_conv3_rw_gpu.py tests/test_step_gpu.py tests/test_bench_config_gpu.py -q -x 2>&1 | tail -3 || exit 1
b() { timeout -k 10 150 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'])"; }
echo "== default"; b
echo "== TECOGAN_PERSIST_WGS=256"; TECOGAN_PERSIST_WGS=256 b
echo "== default cfg4"; timeout -k 10 200 python bench.py --config 4 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-220
echo "== 256 cfg4"; TECOGAN_PERSIST_WGS=256 timeout -k 10 200 python bench.py --config 4 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-220
timeout -k 10 200 python tools/lane_ends.py 2>&1 | tail -7
