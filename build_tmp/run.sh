cd "${GRAFT_REPO_ROOT:?}"
b() { timeout -k 10 150 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'])"; }
for v in 0 50000 200000 0; do echo "== PERSIST_MINPIX=$v"; TECOGAN_PERSIST_MINPIX=$v b; done
