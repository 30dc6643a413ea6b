cd "${GRAFT_REPO_ROOT:?}"
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -q -x 2>&1 | tail -3 || exit 1
bash tools/ab_libs.sh build_tmp/lib_old.so 2>&1 | grep -E "== lib|alone|whole step|bench"
