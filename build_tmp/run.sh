cd "${GRAFT_REPO_ROOT:?}"
b() { timeout -k 10 150 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'])"; }
for v in 0 4 0 4; do echo "== bench RB_TILE=$v"; TECOGAN_RB_TILE=$v b; done
echo "== inference cfg5"; timeout -k 10 200 python tools/bench_inference.py 2>&1 | tail -2
echo "== inference cfg5 RB_TILE=4"; TECOGAN_RB_TILE=4 timeout -k 10 200 python tools/bench_inference.py 2>&1 | tail -2
