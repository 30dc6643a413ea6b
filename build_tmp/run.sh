cd "${GRAFT_REPO_ROOT:?}"
bash tools/ab_libs.sh build_tmp/lib_w4.so 2>&1 | grep -E "== lib|chain alone|whole step|bench"
