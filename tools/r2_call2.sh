#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -q > gpurun_out/c2_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/c2_pytest.log
[ $rc -eq 0 ] || echo "TESTS FAILED (continuing with measurements)"
for v in "TECOGAN_DREAL_BWD=1" "TECOGAN_DREAL_BWD=0" "TECOGAN_RW=0" "TECOGAN_RW=all"; do
  echo "== $v" >> gpurun_out/c2_matrix.log
  env $v timeout -k 10 120 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['final_losses'])" >> gpurun_out/c2_matrix.log 2>&1
done
cat gpurun_out/c2_matrix.log
timeout -k 10 200 python tools/step_breakdown.py > gpurun_out/c2_breakdown.log 2>&1; cat gpurun_out/c2_breakdown.log
timeout -k 10 200 python bench.py --steps 30 --warmup 4 > gpurun_out/c2_bench.json 2> gpurun_out/c2_bench.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/c2_bench.json
