cd "${GRAFT_REPO_ROOT:?}"
for cfg in "0 0" "1 0" "1 4" "1 8"; do set -- $cfg; echo "== FUSED_RESBLOCK_BWD=$1 RB_TILE=$2"
TECOGAN_FUSED_RESBLOCK_BWD=$1 TECOGAN_RB_TILE=$2 timeout -k 10 200 python tools/step_breakdown.py 2>&1 | grep -E "g_bwd alone|whole step"; done
