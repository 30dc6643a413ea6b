cd "${GRAFT_REPO_ROOT:?}"
b() { timeout -k 10 150 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'])"; }
echo "== default"; b
for v in 128 192 256; do echo "== RW_D=$v"; TECOGAN_PERSIST_RW_D=$v b; done
for v in 128 192 224 256; do echo "== RW_G=$v"; TECOGAN_PERSIST_RW_G=$v b; done
for v in 64 80; do echo "== WGS_D=$v RW_D=128"; TECOGAN_PERSIST_WGS_D=$v TECOGAN_PERSIST_RW_D=128 b; done
echo "== default"; b
