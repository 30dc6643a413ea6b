cd "${GRAFT_REPO_ROOT:?}"
b() { timeout -k 10 150 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'])"; }
for v in 0 96 80 64 48 0; do echo "== LDS_BUDGET_KB=$v"; TECOGAN_LDS_BUDGET_KB=$v b; done
TECOGAN_LDS_BUDGET_KB=80 timeout -k 10 200 python tools/step_breakdown.py 2>&1 | grep -E "alone|chain \|\||whole step"
