cd "${GRAFT_REPO_ROOT:?}"
TECOGAN_GBWD_SPLIT=3 timeout -k 10 600 python -m pytest tests/test_bench_config_gpu.py -q -x -k "b4_bf16 or drift" 2>&1 | tail -3 || exit 1
b() { timeout -k 10 150 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'])"; }
for k in 0 1 2 3 4 5 0; do echo "== GBWD_SPLIT=$k"; TECOGAN_GBWD_SPLIT=$k b; done
