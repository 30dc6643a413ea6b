cd "${GRAFT_REPO_ROOT:?}"
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_conv3_rw_gpu.py tests/test_step_gpu.py tests/test_bench_config_gpu.py -q -x 2>&1 | tail -5 || exit 1
b() { timeout -k 10 150 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'])"; }
for r in 1 4 1 4 8 2; do echo "== STATS_REPLICAS=$r"; TECOGAN_STATS_REPLICAS=$r b; done
TECOGAN_STATS_REPLICAS=1 timeout -k 10 200 python tools/step_breakdown.py 2>&1 | grep -E "alone \(lane|whole step"
TECOGAN_STATS_REPLICAS=4 timeout -k 10 200 python tools/step_breakdown.py 2>&1 | grep -E "alone \(lane|whole step"
