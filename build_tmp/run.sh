cd "${GRAFT_REPO_ROOT:?}"
for v in 1 0 1 0; do echo "== RW_128BIG=$v"; TECOGAN_RW_128BIG=$v timeout -k 10 200 python tools/step_breakdown.py 2>&1 | grep -E "g_bwd alone|whole step"; done
