#!/bin/bash
# Kernel-development build that recompiles ONE source with extra flags and links it with the in-tree objects of the others
# (run csrc/build.sh first): hairfastgan_amd/csrc/libhairfast_<name>.so, git-ignored, loaded through HAIRFAST_HIP_LIB.
# Usage: tools/build_one.sh <name> <source without .hip> [hipcc flags, e.g. -DHF_H_TRACE]
set -e
name=$1; src=$2; shift; shift
cd "$(dirname "$0")/../hairfastgan_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
D=/tmp/hf_one_$name
mkdir -p $D
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $src.hip -o $D/$src.o
OBJS=""
for f in api elementwise upfirdn2d style torgb modconv convh convh_enc encoder_ops sean vit gemm_h stem convrow; do
  if [ $f = $src ]; then OBJS="$OBJS $D/$f.o"; else OBJS="$OBJS $f.o"; fi
done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libhairfast_$name.so $OBJS
echo built $(pwd)/libhairfast_$name.so
