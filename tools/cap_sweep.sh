#!/bin/bash
# caps sweep on one box: tools/cap_sweep.sh "G:D:DREAL:FWD ..." [config]   (empty field = default) - bench lines only
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
CFG=${2:-2}
for spec in $1; do
  IFS=: read -r g d dr fw <<< "$spec"
  envs=""
  [ -n "${g:-}" ] && envs="$envs TECOGAN_PERSIST_WGS_G=$g"
  [ -n "${d:-}" ] && envs="$envs TECOGAN_PERSIST_WGS_D=$d"
  [ -n "${dr:-}" ] && envs="$envs TECOGAN_PERSIST_WGS_DREAL=$dr"
  [ -n "${fw:-}" ] && envs="$envs TECOGAN_PERSIST_FWD_G=$fw"
  ms=$(env $envs timeout -k 10 160 python bench.py --config $CFG --steps 40 --warmup 4 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "G=${g:-.} D=${d:-.} DREAL=${dr:-.} FWD=${fw:-.}  $ms ms"
done
