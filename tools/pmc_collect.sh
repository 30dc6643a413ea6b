#!/bin/bash
# HBM traffic per kernel from the PMC counters (run on the GPU box from the repo root):
#   tools/pmc_collect.sh            -> gpurun_out/pmc_{fetch,write}/ and gpurun_out/pmc_summary.json
# FETCH_SIZE and WRITE_SIZE are collected in SEPARATE passes of the same command, with --kernel-trace only
# (MI355X_MICROARCH.md, HBM / rocprofv3 section).  Copy the summary to profiles/ to have it judged.
set -euo pipefail
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT unset)}"
for c in FETCH_SIZE WRITE_SIZE; do
  d=gpurun_out/pmc_$(echo $c | tr 'A-Z' 'a-z' | cut -d_ -f1)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o pmc -- python3 bench.py --steps 2 --warmup 2 --no-roofline --no-cpu-baseline > $d.log 2>&1
done
python3 tools/pmc_summarize.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_summary.json
