#!/bin/bash
# same-box sweep of HIP-runtime environment flags (none is a TECOGAN_* knob): tools/ab_runtime_flags.sh  - bare 40-step bench runs, alternating
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
run() {  # run "NAME=VALUE" (or "base")
  if [ "$1" = base ]; then pre=""; else pre="$1"; fi
  out=$(env $pre timeout -k 10 160 python bench.py --config 2 --steps 40 --warmup 4 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1)
  ms=$(echo "$out" | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null || echo FAILED)
  echo "$1 -> $ms"
}
for rep in 1 2; do
  for s in base DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 base DEBUG_HIP_GRAPH_BATCH_SIZE=1 DEBUG_HIP_GRAPH_BATCH_SIZE=64 \
           GPU_STREAMOPS_CP_WAIT=0 GPU_STREAMOPS_CP_WAIT=1 base ROC_SYSTEM_SCOPE_SIGNAL=0 DEBUG_HIP_DYNAMIC_QUEUES=0 DEBUG_HIP_DYNAMIC_QUEUES=1 \
           HIP_FORCE_DEV_KERNARG=0 DEBUG_HIP_KERNARG_COPY_OPT=0 AMD_OPT_FLUSH=0 base; do
    run "$s"
  done
done
