#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
out=gpurun_out/c6_matrix.log; : > $out
for v in "TECOGAN_CU_RESERVE=0" "TECOGAN_CU_RESERVE=32" "TECOGAN_CU_RESERVE=64" "TECOGAN_CU_RESERVE=96" "TECOGAN_CU_RESERVE=128" "TECOGAN_CU_RESERVE=64 TECOGAN_DREAL_BWD=0" "TECOGAN_CU_RESERVE=0"; do
  echo "== $v" >> $out
  env $v timeout -k 10 120 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['final_losses'])" >> $out 2>&1
done
cat $out
TECOGAN_CU_RESERVE=64 timeout -k 10 200 python tools/step_breakdown.py > gpurun_out/c6_breakdown_r64.log 2>&1; cat gpurun_out/c6_breakdown_r64.log
