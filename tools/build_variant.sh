#!/bin/bash
# Kernel-development build of the library with extra -D flags: hairfastgan_amd/csrc/libhairfast_<name>.so (git-ignored; load it
# through HAIRFAST_HIP_LIB).  Usage: tools/build_variant.sh <name> [hipcc flags, e.g. -DHF_H_SPLIT_STORE16=0]
set -e
name=$1; shift
cd "$(dirname "$0")/../hairfastgan_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
D=/tmp/hf_variant_$name
mkdir -p $D
OBJS=""
for f in api elementwise upfirdn2d style torgb modconv convh convh_enc encoder_ops sean vit gemm_h stem convrow; do
  $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $f.hip -o $D/$f.o &
  OBJS="$OBJS $D/$f.o"
done
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libhairfast_$name.so $OBJS
echo built $(pwd)/libhairfast_$name.so
