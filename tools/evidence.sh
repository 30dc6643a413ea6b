#!/bin/bash
# round-end evidence: full GPU suite, default bench, kernel-trace stats, PMC passes, schedule breakdown, lane overlap, config-4 / inference benches
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
tag=${1:-ev}
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/${tag}_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${tag}_pytest.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/${tag}_smoke.log
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"; cut -c1-200 gpurun_out/${tag}_bench.json
timeout -k 10 300 bash tools/prof_top.sh $tag --steps 13 --warmup 3 --no-cpu-baseline --no-roofline --no-extras > gpurun_out/${tag}_prof_top.log 2>&1; echo "prof rc=$?"
cp gpurun_out/prof_$tag/${tag}_kernel_stats.csv gpurun_out/${tag}_kernel_stats.csv
python tools/overlap_from_trace.py gpurun_out/prof_$tag/${tag}_kernel_trace.csv 3 > gpurun_out/${tag}_overlap_trace.json
timeout -k 10 500 bash tools/pmc_collect.sh $tag gpurun_out/prof_$tag/${tag}_kernel_stats.csv > gpurun_out/${tag}_pmc.log 2>&1; echo "pmc rc=$?"
timeout -k 10 200 python tools/step_breakdown.py > gpurun_out/${tag}_breakdown.log 2>&1; cat gpurun_out/${tag}_breakdown.log
timeout -k 10 200 python tools/lane_ends.py --json gpurun_out/${tag}_lane_overlap.json > gpurun_out/${tag}_lane_ends.log 2>&1; tail -1 gpurun_out/${tag}_lane_ends.log
timeout -k 10 200 python bench.py --config 4 --steps 10 --warmup 3 > gpurun_out/${tag}_bench_cfg4.json 2> /dev/null; cut -c1-200 gpurun_out/${tag}_bench_cfg4.json
timeout -k 10 200 python tools/bench_inference.py > gpurun_out/${tag}_inference_cfg5.json 2>/dev/null; cat gpurun_out/${tag}_inference_cfg5.json
