#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof_top.sh <tag> [bench args]   -> gpurun_out/prof_<tag>/ + top kernels
set -euo pipefail
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT unset)}"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o $tag -- python3 bench.py "$@" > gpurun_out/prof_$tag.log 2>&1
python3 - "$tag" <<'PY'
import csv, glob, sys
f = glob.glob(f"gpurun_out/prof_{sys.argv[1]}/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print(r["Name"][:100], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
tail -1 gpurun_out/prof_$tag.log | cut -c1-200
