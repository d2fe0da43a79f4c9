#!/bin/bash
# Builds libhairfast_hip.so in-tree for gfx950 (MI355X).  hipcc cross-compiles
# without a GPU.  Usage: hairfastgan_amd/csrc/build.sh [extra hipcc flags]
# Out-of-date objects are compiled concurrently (convh.hip alone takes ~1 min).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC $*"
OBJS=""
PIDS=""
for f in api elementwise upfirdn2d style torgb modconv convh convh_enc encoder_ops sean vit gemm_h stem convrow; do
  if [ ! -f $f.o ] || [ $f.hip -nt $f.o ] || [ hf_common.h -nt $f.o ] || [ conv_common.h -nt $f.o ] || [ ../../include/hairfast_hip.h -nt $f.o ]; then
    $HIPCC $FLAGS -c $f.hip -o $f.o &
    PIDS="$PIDS $!"
  fi
  OBJS="$OBJS $f.o"
done
for p in $PIDS; do
  wait $p  # set -e: a failed compile aborts the build
done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libhairfast_hip.so $OBJS
echo built $(pwd)/libhairfast_hip.so
