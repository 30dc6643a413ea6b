#!/bin/bash
# per-kernel times of the config-5 inference loop under rocprofv3 (on the GPU box): tools/prof_inference.sh <tag>
set -euo pipefail
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o $tag -- python3 tools/bench_inference.py > gpurun_out/prof_$tag.log 2>&1
python3 - "$tag" <<'PY'
import csv, glob, sys
f = glob.glob(f"gpurun_out/prof_{sys.argv[1]}/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
frames = 13 * 120.0   # 1 + 2 + 10 sequences of 120 frames
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print(f'{r["Name"][:86]:86s} calls/frame {int(r["Calls"]) / frames:5.2f} avg {float(r["AverageNs"]) / 1e3:7.2f} us  per-frame {float(r["TotalDurationNs"]) / frames / 1e3:6.1f} us')
print("kernel time per frame", tot / frames / 1e3, "us")
PY
tail -1 gpurun_out/prof_$tag.log
