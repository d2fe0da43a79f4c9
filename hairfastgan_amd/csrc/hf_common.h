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

// Async 16-byte-per-lane global -> LDS copy (global_load_lds_dwordx4).  The LDS
// destination is wave-uniform base + lane*16 (the base is taken from the first
// active lane); the global source is per lane.  Completion is tracked by vmcnt:
// hipcc drains it before the next __syncthreads().
#ifndef HF_GLDS16_DEFINED
#define HF_GLDS16_DEFINED
__device__ __forceinline__ void hf_glds16(const float *gsrc_lane, float *lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc_lane,
                                   (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}
// 4-byte-per-lane variant (global_load_lds_dword): LDS destination = base + lane*4.
__device__ __forceinline__ void hf_glds4(const float *gsrc_lane, float *lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc_lane,
                                   (__attribute__((address_space(3))) void *)lds_wave_base, 4, 0, 0);
}
// Lane-masked forms: only lanes with `active` copy (the others' LDS slots keep their
// contents).  Separate helpers so that the CPU kernel interpreter can model the EXEC mask.
__device__ __forceinline__ void hf_glds16_if(bool active, const float *gsrc_lane, float *lds_wave_base) {
  if (active) hf_glds16(gsrc_lane, lds_wave_base);
}
__device__ __forceinline__ void hf_glds4_if(bool active, const float *gsrc_lane, float *lds_wave_base) {
  if (active) hf_glds4(gsrc_lane, lds_wave_base);
}
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
