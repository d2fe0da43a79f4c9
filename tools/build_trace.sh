#!/bin/bash
# Kernel-development build with the wave-timeline probes of csrc/convh.hip (HF_H_TRACE): hairfastgan_amd/csrc/libhairfast_trace.so
# (git-ignored; load it through HAIRFAST_HIP_LIB, e.g. tools/probes/trace_layer.py).
set -e
cd "$(dirname "$0")/../hairfastgan_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
mkdir -p /tmp/hf_trace_objs
OBJS=""
for f in api elementwise upfirdn2d style torgb modconv convh convh_enc encoder_ops sean vit gemm_h stem convrow; do
  $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DHF_H_TRACE $* -c $f.hip -o /tmp/hf_trace_objs/$f.o &
  OBJS="$OBJS /tmp/hf_trace_objs/$f.o"
done
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libhairfast_trace.so $OBJS
echo built $(pwd)/libhairfast_trace.so
