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

// Pins a float to its fp32-rounded value (an empty asm the optimiser cannot see through):
// stops instruction selection from fusing a preceding multiply into a following fp16
// conversion (v_fma_mixlo_f16 rounds the UNROUNDED product).
#ifndef HF_OPAQUE_F32
#define HF_OPAQUE_F32(v) asm volatile("" : "+v"(v))
// same for an int: recomputed where it is used instead of being hoisted out of the loop
// into a register that stays live (address arithmetic of software-pipelined loads)
#define HF_OPAQUE_I32(v) asm volatile("" : "+v"(v))
#endif

// Hand-scheduled LDS-DMA pipeline primitives.  hipcc models global_load_lds as a FLAT access
// to both address spaces ("pending flat"): while one is in flight EVERY wait it inserts is
// vmcnt(0) / lgkmcnt(0), which serialises LDS fragment prefetch and register prefetch behind
// the DMA.  Issuing the DMA from inline asm hides it from that bookkeeping; completion is then
// the kernel's job: hf_barrier_keep_young<N>() before the data is read.
//   hf_glds16_raw: same contract as hf_glds16 (LDS dst = wave-uniform base + lane*16).
//   hf_barrier_keep_young<N>: workgroup barrier that drains LDS traffic and all but the
//     youngest N vector-memory operations of the wave.  N > 0 is only sound when the operations
//     that must have landed and the young ones are of the SAME kind: register loads retire in
//     order among themselves, but LDS-DMA copies retire through the LDS path and are not
//     ordered against register loads (measured: stale DMA data with N = the number of younger
//     register loads).  With DMA copies pending use N = 0.
//   M0: the asm writes M0 without listing it as a clobber - the AMDGPU backend treats M0 as a
//     RESERVED register (hipcc rejects the clobber with -Winline-asm "reserved registers: m0"): it
//     never keeps a value live in M0 across statements, it re-materialises M0 immediately before
//     each of its own instructions that read it.
#ifndef HF_BARRIER_KEEP_DEFINED
#define HF_BARRIER_KEEP_DEFINED
__device__ __forceinline__ void hf_glds16_raw(const float *gsrc_lane, float *lds_wave_base) {
  const unsigned base = __builtin_amdgcn_readfirstlane(
      (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) void *)lds_wave_base);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(base), "v"(gsrc_lane) : "memory");
}
// LDS byte address of a pointer into the dynamic LDS region (what M0 takes)
__device__ __forceinline__ unsigned hf_lds_addr(const void *lds_ptr) {
  return (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) const void *)lds_ptr;
}
// wave-uniform 64-bit base (SGPR pair) + 32-bit per-lane byte offset: one VGPR of address.
// The LDS destination is an integer byte address (hf_lds_addr of the region base + offset):
// LDS *pointers* selected at run time degrade to generic pointers with aperture checks.
__device__ __forceinline__ void hf_glds16_raw_s(const void *gsrc_uniform, unsigned lane_byte_offset,
                                                unsigned lds_wave_addr) {
  const unsigned base = __builtin_amdgcn_readfirstlane(lds_wave_addr);
  // readfirstlane: the pointer is wave-uniform by contract, but the compiler may hold it in
  // VGPRs (e.g. selected by a wave index) - an "s" operand needs a provably scalar value
  const unsigned long long g = (unsigned long long)gsrc_uniform;
  const unsigned g_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(g >> 32));
  const unsigned g_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)g);  // unsigned: no sign extension
  const unsigned long long gs = ((unsigned long long)g_hi << 32) | g_lo;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(base), "v"(lane_byte_offset), "s"(gs)
               : "memory");
}
// lane-masked form: lanes with `active` false copy nothing (their LDS slots keep their contents)
__device__ __forceinline__ void hf_glds16_raw_s_if(bool active, const void *gsrc_uniform, unsigned lane_byte_offset,
                                                   unsigned lds_wave_addr) {
  if (active) hf_glds16_raw_s(gsrc_uniform, lane_byte_offset, lds_wave_addr);
}
// workgroup barrier that only orders LDS traffic (vector-memory operations stay in flight)
__device__ __forceinline__ void hf_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int NYOUNG>
__device__ __forceinline__ void hf_barrier_keep_young() {
  static_assert(NYOUNG >= 0 && NYOUNG < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NYOUNG) : "memory");
}
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

// ---- fp32 -> fp16 (hi, lo) operand split of the f16x3 / f16 matrix-core modes ----
// hi = fp16(v), lo = fp16(v - hi): 22 significant bits while lo stays a normal fp16 number, i.e.
// for |v| >= 2^-3; below that the pair carries an ABSOLUTE error of at most 2^-25 (half the
// fp16 subnormal step), which is why producers pre-scale by powers of two so that the large
// operands of a dot product sit well inside [2^-2, 2^15] (weights: hf_conv_split_weights_f16;
// modulation: hf_style_normalize_f32).  Range: both parts SATURATE at +-65504 instead of turning
// into inf (inf - inf = NaN in the accumulator), so the pair is exact-ish up to 131008 and
// clamps beyond; every clamped element is counted in hf_f16_overflow (hf_f16_overflow_count()),
// so a caller can detect that a tensor left the representable range and re-run in f32 mode.
// Translation units that split define HF_WANT_F16_SPLIT before including this header; each gets
// its own counter (no relocatable device code), hf_f16_overflow_count() sums them.
#ifdef HF_WANT_F16_SPLIT
#define HF_F16_MAX 65504.0f
static __device__ unsigned int hf_f16_overflow_dev;
static inline unsigned int hf_f16_overflow_read_tu(int reset) {  // synchronous (diagnostics only)
  unsigned int v = 0, z = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(hf_f16_overflow_dev), sizeof(v)) != hipSuccess) return 0xffffffffu;
  if (reset) (void)hipMemcpyToSymbol(HIP_SYMBOL(hf_f16_overflow_dev), &z, sizeof(z));
  return v;
}
__device__ __forceinline__ void hf_split_f16(float v, _Float16 &hi, _Float16 &lo, bool &ovf) {
  // hi and lo must both derive from the fp32-ROUNDED v: left alone, hipcc stores
  // hi = fp16(fp32(x*s)) but subtracts v_fma_mixlo_f16's fp16(x*s unrounded); at an fp16
  // rounding tie of the fp32 product the two differ by one fp16 ulp (seen on hardware).
  HF_OPAQUE_F32(v);
  ovf = ovf || !(fabsf(v) <= 2.0f * HF_F16_MAX);  // also true for NaN
  const float vh = fminf(fmaxf(v, -HF_F16_MAX), HF_F16_MAX);
  hi = (_Float16)vh;
  const float r = v - (float)hi;
  lo = (_Float16)fminf(fmaxf(r, -HF_F16_MAX), HF_F16_MAX);
}
__device__ __forceinline__ void hf_note_overflow(bool ovf) {
  if (ovf) atomicAdd(&hf_f16_overflow_dev, 1u);
}
// Split of FOUR values with one range test for the whole wave: the largest |v| of the four is compared with the
// fp16 maximum and the wave votes; in range (the normal case) the split needs no clamps and no per-element tests
// (4 instead of 9 VALU instructions per element), otherwise the saturating element-wise form above runs.  Both
// forms give identical bits for values within +-65504.  Convergent: every lane of the wave must call it.
typedef _Float16 hf_half4 __attribute__((ext_vector_type(4)));
#ifndef HF_WAVE_ANY_DEFINED
#define HF_WAVE_ANY_DEFINED
__device__ __forceinline__ bool hf_wave_any(bool p) { return __any((int)p) != 0; }
#endif
__device__ __forceinline__ void hf_split4_f16(const float (&vin)[4], hf_half4 &h, hf_half4 &l, bool &ovf) {
  float v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v[k] = vin[k];
    HF_OPAQUE_F32(v[k]);  // hi and lo from the same fp32-rounded value (see hf_split_f16)
  }
  const float m = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
  if (hf_wave_any(!(m <= HF_F16_MAX))) {  // also NaN
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      _Float16 hv, lv;
      hf_split_f16(v[k], hv, lv, ovf);
      h[k] = hv;
      l[k] = lv;
    }
  } else {
    // on channel PAIRS: v_cvt_pk_f16_f32, two v_cvt_f32_f16 (the upper half by SDWA), v_pk_add_f32, v_cvt_pk_f16_f32 - 2.5 VALU
    // issues per element and the halves already packed (element-wise hipcc spent 5.5 incl. v_bfi / v_perm packing; same bits:
    // the packed conversion rounds to nearest even like the scalar one)
    typedef _Float16 hf_half2_ __attribute__((ext_vector_type(2)));
    typedef float hf_float2_ __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      hf_float2_ x;
      x.x = v[2 * p];
      x.y = v[2 * p + 1];
      const hf_half2_ hh = __builtin_convertvector(x, hf_half2_);
      const hf_float2_ r = x - __builtin_convertvector(hh, hf_float2_);
      const hf_half2_ ll = __builtin_convertvector(r, hf_half2_);
      h[2 * p] = hh.x;
      h[2 * p + 1] = hh.y;
      l[2 * p] = ll.x;
      l[2 * p + 1] = ll.y;
    }
  }
}
#endif  // HF_WANT_F16_SPLIT

// Value of the neighbouring lane (lane-1 / lane+1) by a DPP wavefront shift: one VALU instruction, no LDS
// round trip (the __shfl_* forms are ds_bpermute with an address register and ~100 cycles of latency each).
// Lane 0 (up) / lane 63 (down) receive 0 (bound_ctrl: an out-of-range source lane reads as 0 - with the `old` operand form instead,
// hipcc initialises the destination with a v_mov_b32 0 in front of every shift: 1.5 extra VALU issues per output of the fused
// upsampling epilogue, which is bound by VALU issue - tools/probes/valu_rate.hip).
#ifndef HF_LANE_SHIFT_DEFINED
#define HF_LANE_SHIFT_DEFINED
__device__ __forceinline__ float hf_lane_up(float v) {  // from lane-1: DPP wave_shr:1 (0x138)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float hf_lane_down(float v) {  // from lane+1: DPP wave_shl:1 (0x130)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
#endif

// The two half-waves exchange one dword each (v_permlane32_swap_b32, gfx950): afterwards lanes 0-31 hold (a of lane i,
// a of lane i+32) and lanes 32-63 hold (b of lane i-32, b of lane i) in (a, b).  Used to turn two 8-byte stores of a
// split activation (a lane owns HALF of a 16-byte hi unit and half of a lo unit) into one 16-byte store per lane: the low
// half-wave writes whole hi units, the high half-wave whole lo units.
#ifndef HF_HALF_SWAP_DEFINED
#define HF_HALF_SWAP_DEFINED
__device__ __forceinline__ void hf_half_swap(unsigned &a, unsigned &b) {
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0];
  b = r[1];
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
