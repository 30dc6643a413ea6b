cd "${GRAFT_REPO_ROOT:?}"
for i in 1 2; do timeout -k 10 900 python -m pytest tests -m gpu -q 2>&1 | tail -2; done
