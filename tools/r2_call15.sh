#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
for v in "TECOGAN_DREAL_BWD=1" "TECOGAN_DREAL_BWD=0" "TECOGAN_FUSED_RESBLOCK_BWD=1" "TECOGAN_RW=all" "TECOGAN_CU_RESERVE=96" "TECOGAN_DREAL_BWD=1"; do
  echo "== $v"
  env $v timeout -k 10 120 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['final_losses'])"
done
