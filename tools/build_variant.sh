#!/bin/bash
# A/B and diagnostic builds: tools/build_variant.sh <name> <source stem>[,<stem>...] [hipcc flags]
#   -> _ab/libtecogan_hip_<name>.so = the default objects with the named sources recompiled under the given flags
# (the default library and its objects are untouched; load with TECOGAN_LIB=$PWD/_ab/libtecogan_hip_<name>.so)
set -euo pipefail
cd "$(dirname "$0")/../pytorch-tecogan_amd/csrc"
name=$1; stems=$2; shift 2
out=../../_ab; mkdir -p $out/$name
objs=""
for f in conv_mfma wgrad_mfma wgrad_group warp elementwise fnet resblock resblock_ws convt_mfma convt_cw conv4s2_mfma conv4s2d_cw conv_s2_cw d_tail runtime conv3_rw conv3_cw vgg conv_rgb rgb_bwd; do
  if [[ ",$stems," == *",$f,"* ]]; then
    per_file=""; [ $f = conv3_rw ] && per_file="-fno-slp-vectorize"   # (as csrc/build.sh)
    [ -z "${NO_PRELOAD:-}" ] && per_file="$per_file -mllvm -amdgpu-kernarg-preload-count=16"   # (as csrc/build.sh's FLAGS)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $per_file "$@" -c $f.hip -o $out/$name/$f.o
    objs="$objs $out/$name/$f.o"
  else
    objs="$objs $f.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libtecogan_hip_$name.so $objs
echo "built _ab/libtecogan_hip_$name.so"
