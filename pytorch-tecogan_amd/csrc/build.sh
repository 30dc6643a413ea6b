#!/bin/bash
# Builds libtecogan_hip.so for gfx950 in-tree (next to this script).  Usage: build.sh [extra hipcc flags]
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
pids=""
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
for f in conv_mfma wgrad_mfma wgrad_group warp elementwise fnet resblock convt_mfma conv4s2_mfma runtime conv3_rw vgg conv_rgb rgb_bwd resblock2; do
  if [ ! -f $f.o ] || [ $f.hip -nt $f.o ] || [ common.h -nt $f.o ] || [ ../../include/tecogan_hip.h -nt $f.o ]; then
    rm -f $f.o   # a failed compile must not leave the previous object behind for the link below
    $HIPCC $FLAGS "$@" -c $f.hip -o $f.o &
    pids="$pids $!"
  fi
done
for p in $pids; do wait $p; done   # (a bare `wait` returns 0 whatever the jobs did)
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libtecogan_hip.so conv_mfma.o wgrad_mfma.o wgrad_group.o warp.o elementwise.o fnet.o resblock.o convt_mfma.o conv4s2_mfma.o runtime.o conv3_rw.o vgg.o conv_rgb.o rgb_bwd.o resblock2.o
echo "built $(pwd)/libtecogan_hip.so"
