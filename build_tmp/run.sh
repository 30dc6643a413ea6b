cd "${GRAFT_REPO_ROOT:?}"
b() { timeout -k 10 150 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'])"; }
echo "== default"; b
echo "== DREAL_BWD=0"; TECOGAN_DREAL_BWD=0 b
echo "== STATS_REPLICAS=8"; TECOGAN_STATS_REPLICAS=8 b
echo "== FUSED_RESBLOCK_BWD=1"; TECOGAN_FUSED_RESBLOCK_BWD=1 b
echo "== RW=all"; TECOGAN_RW=all b
echo "== RW=0"; TECOGAN_RW=0 b
echo "== RB_TILE=8"; TECOGAN_RB_TILE=8 b
echo "== default"; b
