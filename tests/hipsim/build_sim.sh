#!/bin/bash
# TEST INFRASTRUCTURE - builds tests/hipsim/libhairfast_sim.so: the SAME kernel
# sources as the product, compiled as host C++ against the hipsim shim (concurrently).
set -e
cd "$(dirname "$0")"
CXX=${HIPSIM_CXX:-/opt/rocm/lib/llvm/bin/clang++}
SRC=../../hairfastgan_amd/csrc
OBJS=""
PIDS=""
for f in api elementwise upfirdn2d style torgb modconv convh convh_enc encoder_ops sean vit gemm_h stem convrow; do
  $CXX -x c++ -std=c++17 -O2 -fPIC -Wno-psabi -Wno-pass-failed -I. -c $SRC/$f.hip -o /tmp/hipsim_$f.o &
  PIDS="$PIDS $!"
  OBJS="$OBJS /tmp/hipsim_$f.o"
done
$CXX -std=c++17 -O2 -fPIC -Wno-psabi -I. -c hipsim.cpp -o /tmp/hipsim_rt.o &
PIDS="$PIDS $!"
for p in $PIDS; do
  wait $p
done
$CXX -shared -o libhairfast_sim.so $OBJS /tmp/hipsim_rt.o
echo built $(pwd)/libhairfast_sim.so
