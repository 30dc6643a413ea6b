cd "${GRAFT_REPO_ROOT:?}"
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "batchnorm" 2>&1 | tail -3 || exit 1
for lib in "$PWD/build_tmp/lib_old.so" "" "$PWD/build_tmp/lib_old.so" ""; do echo "== lib=${lib:-<shipped>}"
TECOGAN_LIB=$lib timeout -k 10 120 python bench.py --steps 40 --warmup 4 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); h=d['roofline']['hbm_kernels']; print('bench', d['ms_per_step'], {k[:14]: v['avg_launch_us'] for k,v in h.items() if k.startswith('bn_')})"; done
