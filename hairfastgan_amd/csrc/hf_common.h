// hf_common.h - shared helpers for the gfx950 kernels of libhairfast_hip.so.
// CDNA4 only: 64-wide wavefronts, LDS carved from one dynamic region.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/hairfast_hip.h"

#define HF_WAVE 64

// All LDS comes from the single dynamic region (guide G17: no static
// __shared__ in front of it, 16-byte aligned carve offsets).
#ifndef HF_DYN_LDS
#define HF_DYN_LDS extern __shared__ __attribute__((aligned(16))) unsigned char hf_dyn_lds[]
#endif

static inline int hf_launch_status() {
  return hipGetLastError() == hipSuccess ? HF_OK : HF_E_LAUNCH;
}

static inline int hf_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float hf_lrelu(float v, float alpha, float scale) {
  return (v > 0.0f ? v : v * alpha) * scale;
}

// Sum across the 64 lanes of a wave; every lane gets the total.
__device__ __forceinline__ float hf_wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, HF_WAVE);
  return v;
}
