cd "${GRAFT_REPO_ROOT:?}"
b() { timeout -k 10 150 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'], d['final_losses'])"; }
echo "== default"; b
for v in 64 96 128 160; do echo "== CHAIN_MASK CU_RESERVE=$v"; TECOGAN_CHAIN_MASK=1 TECOGAN_CU_RESERVE=$v b; done
echo "== default"; b
