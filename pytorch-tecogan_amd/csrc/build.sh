#!/bin/bash
# Builds libtecogan_hip.so for gfx950 in-tree (next to this script).  Usage: build.sh [--experiments] [extra hipcc flags]
#   --experiments   also compile the variants that were built, measured slower and rejected (-DTG_EXPERIMENTS: exp/resblock2.hip, exp/resblock2_ws.hip, exp/resblock_pp.hip,
#                   the 64 x 128 work-list blocks, tg_bn_bwd_fused, tg_conv's stats_mode 3, the item-walking fold, the forced
#                   trunk tile) into libtecogan_hip_experiments.so - objects under exp/, the default library is untouched.
#                   Load it with TECOGAN_LIB=.../libtecogan_hip_experiments.so; tests: pytest -m experiments.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
SRCS="conv_mfma wgrad_mfma wgrad_group warp elementwise fnet resblock resblock_ws convt_mfma convt_cw conv4s2_mfma conv4s2d_cw conv_s2_cw d_tail runtime conv3_rw conv3_cw vgg conv_rgb rgb_bwd"
OUT=libtecogan_hip.so
OBJ=.
EXTRA=""
if [ "${1:-}" = "--experiments" ]; then
  shift
  SRCS="$SRCS resblock2 resblock2_ws resblock_pp"
  OUT=libtecogan_hip_experiments.so
  OBJ=exp
  EXTRA="-DTG_EXPERIMENTS"
  mkdir -p exp
fi
pids=""
# -amdgpu-kernarg-preload-count: kernels whose arguments are passed one by one (everything but the by-value launch structs) get their
# first 16 dwords preloaded into SGPRs with the wave - no s_load round trip in front of a workgroup's first address (r05_k log)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-kernarg-preload-count=16 $EXTRA"
objs=""
for f in $SRCS; do
  o=$OBJ/$f.o
  objs="$objs $o"
  src=$f.hip; [ -f $src ] || src=exp/$f.hip   # (the rejected variants' sources live under exp/, next to the experiments build's objects)
  if [ ! -f $o ] || [ $src -nt $o ] || [ common.h -nt $o ] || [ rbw_common.h -nt $o ] || [ ../../include/tecogan_hip.h -nt $o ]; then
    rm -f $o   # a failed compile must not leave the previous object behind for the link below
    # conv3_rw: the producer waves' epilogue arithmetic shares a SIMD with the consumer's MFMA stream; SLP-packed f32 operations
    # (v_pk_add_f32 / v_pk_mul_f32) cost ~+25 cycles each beside MFMAs (MI355X_MICROARCH.md, 'price of one filler')
    per_file=""; [ $f = conv3_rw ] && per_file="-fno-slp-vectorize"
    $HIPCC $FLAGS -I. $per_file "$@" -c $src -o $o &
    pids="$pids $!"
  fi
done
for p in $pids; do wait $p; done   # (a bare `wait` returns 0 whatever the jobs did)
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT $objs
echo "built $(pwd)/$OUT"
