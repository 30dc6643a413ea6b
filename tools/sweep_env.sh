#!/bin/bash
# same-box sweep of environment settings: tools/sweep_env.sh <steps> "<ENV=.. ENV=..>" "<...>" ...   ("" = defaults)
# prints bench.py's ms_per_step for each setting (no CPU baseline, no roofline pass, no extras)
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
steps=$1; shift
for cfg in "$@"; do
  ms=$(env $cfg timeout -k 10 150 python bench.py --steps $steps --warmup 4 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  printf "%-90s %s\n" "${cfg:-<defaults>}" "$ms"
done
