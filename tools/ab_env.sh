#!/bin/bash
# same-box A/B of one TECOGAN_* knob: tools/ab_env.sh VAR "valA valB" [configs "2 4"]  - alternating runs, bench lines + the step's breakdown
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
VAR=$1; VALS=${2:-"0 1"}; CFGS=${3:-2}
for rep in 1 2; do
  for v in $VALS; do
    echo "== $VAR=$v (run $rep)"
    if [ $rep = 1 ]; then env $VAR=$v timeout -k 10 200 python tools/step_breakdown.py 2>&1 | grep -E "alone \(lane|g_bwd alone|d_real alone|d_fake alone|d_fake_bwd alone|chain alone|whole step"; fi
    for c in $CFGS; do
      env $VAR=$v timeout -k 10 160 python bench.py --config $c --steps 40 --warmup 4 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench config', $c, d['ms_per_step'])"
    done
  done
done
