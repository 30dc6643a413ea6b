#!/bin/bash
# Per-kernel hardware counters of the benchmarked step (run on the GPU box from the repo root):
#   tools/pmc_collect.sh [tag]     -> gpurun_out/pmc_{fetch,write,mfma}/ and gpurun_out/<tag>_pmc_summary.json
# FETCH_SIZE, WRITE_SIZE and the SQ/GRBM counters are collected in SEPARATE passes of the same command, with --kernel-trace
# only (MI355X_MICROARCH.md, 'HBM' and 'rocprofv3 PMC slots').  Copy the summary to profiles/ to have it judged.
set -euo pipefail
tag=${1:-pmc}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT unset)}"
pass() {  # dir, counters...
  local d=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $d -o pmc -- python3 bench.py --steps 2 --warmup 2 --no-roofline --no-cpu-baseline --no-extras > $d.log 2>&1
}
pass gpurun_out/pmc_fetch FETCH_SIZE
pass gpurun_out/pmc_write WRITE_SIZE
pass gpurun_out/pmc_mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
python3 tools/pmc_summarize.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/${tag}_pmc_summary.json gpurun_out/pmc_mfma ${2:-}
