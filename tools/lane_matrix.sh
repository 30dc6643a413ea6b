#!/bin/bash
# bench.py under the schedule switches of step.TecoGANStep (one JSON line each -> gpurun_out/lane_matrix.log)
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?run through gpurun}"
mkdir -p gpurun_out
out=gpurun_out/lane_matrix.log
: > $out
run() {
  echo "== $*" >> $out
  env "$@" python bench.py --steps 30 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['final_losses'])" >> $out 2>&1 || echo failed >> $out
}
run TECOGAN_LANES=0
run TECOGAN_LANES=1 TECOGAN_CU_RESERVE=0 TECOGAN_DREAL_BWD=0
run TECOGAN_LANES=1 TECOGAN_CU_RESERVE=0 TECOGAN_DREAL_BWD=1
run TECOGAN_LANES=1 TECOGAN_CU_RESERVE=64 TECOGAN_DREAL_BWD=0
run TECOGAN_LANES=1 TECOGAN_CU_RESERVE=64 TECOGAN_DREAL_BWD=1
run TECOGAN_LANES=1 TECOGAN_CU_RESERVE=32 TECOGAN_DREAL_BWD=1
run TECOGAN_LANES=1 TECOGAN_CU_RESERVE=96 TECOGAN_DREAL_BWD=1
cat $out
