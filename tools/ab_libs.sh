#!/bin/bash
# same-box A/B of two library builds: tools/ab_libs.sh <other .so> [tests to run first]
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OTHER=$PWD/$1; shift
if [ $# -gt 0 ]; then timeout -k 10 600 python -m pytest "$@" -q 2>&1 | tail -3; fi
for lib in "$OTHER" "" "$OTHER" ""; do
  echo "== lib=${lib:-<shipped>}"
  TECOGAN_LIB=$lib timeout -k 10 200 python tools/step_breakdown.py 2>&1 | grep -E "alone \(lane|g_bwd alone|chain alone|whole step"
  TECOGAN_LIB=$lib timeout -k 10 120 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'])"
done
