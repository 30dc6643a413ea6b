#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
timeout -k 10 200 python bench.py --steps 40 --warmup 4 --no-cpu-baseline > gpurun_out/c8_bench.json 2> gpurun_out/c8_bench.err; echo "bench rc=$?"; grep "timed region" gpurun_out/c8_bench.err
timeout -k 10 200 python tools/step_breakdown.py > gpurun_out/c8_breakdown.log 2>&1; cat gpurun_out/c8_breakdown.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c8_bench.json').read().strip().splitlines()[-1])
for k,v in d['roofline']['families'].items(): print(f"{k:66s} {v['launches']:4d} {v['ms']:.3f} ms {v['tflops']:7.1f} TF/s")
PY
