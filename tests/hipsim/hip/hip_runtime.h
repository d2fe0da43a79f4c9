// TEST INFRASTRUCTURE - "hipsim": a tiny lock-step CPU interpreter for the HIP
// kernel sources of hairfastgan_amd/csrc, used ONLY by the CPU test-suite
// (tests/test_sim_*.py) to check kernel index arithmetic, LDS layouts, MFMA
// fragment maps and barriers in a container that has no GPU.  It shadows
// <hip/hip_runtime.h> when the .hip files are compiled as plain C++ by host
// clang; the product library is never built against it and nothing here ships.
//
// Model: every thread of a block is a ucontext fiber on one OS thread;
// __syncthreads and wave-wide ops (shuffles, v_mfma_f32_32x32x2_f32) are
// rendezvous points resolved by a round-robin scheduler.  MFMA uses the
// documented gfx950 fragment maps (cdna_hip_programming.md section 3):
//   A[i][k] = lane(i + 32k).a,  B[k][j] = lane(j + 32k).b,
//   D row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31, fmaf chain over k.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <functional>

namespace hipsim {
struct Dim3 {
  unsigned x, y, z;
  Dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
extern Dim3 threadIdx_, blockIdx_, blockDim_, gridDim_;
typedef float f32x16 __attribute__((ext_vector_type(16)));
void syncthreads();
float shfl_xor(float v, int mask);
float shfl_rel(float v, int delta);  // value of lane (lane + delta), own value if out of the wave
float shfl_rel0(float v, int delta);  // same, 0 if out of the wave (DPP wavefront shifts)
f32x16 mfma32x32x2(float a, float b, f32x16 c);
f32x16 mfma32x32x16h(const float *a8, const float *b8, f32x16 c);
void glds16(const float *gsrc_lane, float *lds_wave_base);
void glds4(const float *gsrc_lane, float *lds_wave_base);
void glds_masked(bool active, int bytes, const float *gsrc_lane, float *lds_wave_base);
void launch(const std::function<void()> &body, Dim3 grid, Dim3 block, size_t shmem);
unsigned char *dyn_lds();
}  // namespace hipsim

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__
#define threadIdx ::hipsim::threadIdx_
#define blockIdx ::hipsim::blockIdx_
#define blockDim ::hipsim::blockDim_
#define gridDim ::hipsim::gridDim_
#define HF_DYN_LDS unsigned char *hf_dyn_lds = ::hipsim::dyn_lds()

typedef ::hipsim::Dim3 dim3;
typedef void *hipStream_t;
typedef int hipError_t;
static const hipError_t hipSuccess = 0;
inline hipError_t hipGetLastError() { return hipSuccess; }

struct alignas(16) float4 {
  float x, y, z, w;
};
struct alignas(16) int4 {
  int x, y, z, w;
};
struct alignas(8) float2 {
  float x, y;
};
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

inline void __syncthreads() { ::hipsim::syncthreads(); }
inline float __shfl_xor(float v, int mask, int /*width*/ = 64) { return ::hipsim::shfl_xor(v, mask); }
inline float __shfl_up(float v, int d, int /*width*/ = 64) { return ::hipsim::shfl_rel(v, -d); }
inline float __shfl_down(float v, int d, int /*width*/ = 64) { return ::hipsim::shfl_rel(v, d); }
inline float rsqrtf(float v) { return 1.0f / sqrtf(v); }
// single OS thread, fibers only switch at rendezvous points: plain read-modify-write is atomic
inline unsigned int atomicAdd(unsigned int *p, unsigned int v) { unsigned int o = *p; *p = o + v; return o; }
inline void __threadfence() {}
inline unsigned int atomicMax(unsigned int *p, unsigned int v) { unsigned int o = *p; if (v > o) *p = v; return o; }
inline unsigned int __float_as_uint(float f) { unsigned int u; __builtin_memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned int u) { float f; __builtin_memcpy(&f, &u, 4); return f; }
#define HIP_SYMBOL(x) (&(x))
inline hipError_t hipMemcpyFromSymbol(void *dst, const void *sym, size_t n) { __builtin_memcpy(dst, sym, n); return hipSuccess; }
inline hipError_t hipMemcpyToSymbol(void *sym, const void *src, size_t n) { __builtin_memcpy(sym, src, n); return hipSuccess; }
template <class T> inline T min(T a, T b) { return a < b ? a : b; }
template <class T> inline T max(T a, T b) { return a > b ? a : b; }

#define HF_GLDS16_DEFINED
inline void hf_glds16(const float *gsrc_lane, float *lds_wave_base) { ::hipsim::glds16(gsrc_lane, lds_wave_base); }
inline void hf_glds4(const float *gsrc_lane, float *lds_wave_base) { ::hipsim::glds4(gsrc_lane, lds_wave_base); }
inline void hf_glds16_if(bool a, const float *g, float *l) { ::hipsim::glds_masked(a, 16, g, l); }
inline void hf_glds4_if(bool a, const float *g, float *l) { ::hipsim::glds_masked(a, 4, g, l); }

#define HF_WAVE_ANY_DEFINED
inline bool hf_wave_any(bool p) {  // wave vote through the shuffle rendezvous
  float v = p ? 1.0f : 0.0f;
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, ::hipsim::shfl_xor(v, m));
  return v > 0.0f;
}
#define HF_HALF_SWAP_DEFINED
inline void hf_half_swap(unsigned &a, unsigned &b) {  // v_permlane32_swap_b32: a[32:63] <-> b[0:31]
  const float pa = ::hipsim::shfl_xor(__builtin_bit_cast(float, a), 32), pb = ::hipsim::shfl_xor(__builtin_bit_cast(float, b), 32);
  if ((threadIdx.x & 63) < 32) b = __builtin_bit_cast(unsigned, pa);
  else a = __builtin_bit_cast(unsigned, pb);
}
#define HF_LANE_SHIFT_DEFINED
inline float hf_lane_up(float v) { return ::hipsim::shfl_rel0(v, -1); }
inline float hf_lane_down(float v) { return ::hipsim::shfl_rel0(v, 1); }
#define HF_OPAQUE_F32(v) ((void)0)
#define HF_OPAQUE_I32(v) ((void)0)
#define HF_BARRIER_KEEP_DEFINED
template <int NYOUNG> inline void hf_barrier_keep_young() { ::hipsim::syncthreads(); }
inline void hf_barrier_lds() { ::hipsim::syncthreads(); }
inline void hf_glds16_raw(const float *gsrc_lane, float *lds_wave_base) { ::hipsim::glds16(gsrc_lane, lds_wave_base); }
inline void hf_glds16_raw_s_if(bool active, const void *g, unsigned off, unsigned lds_addr) {
  ::hipsim::glds_masked(active, 16, reinterpret_cast<const float *>(static_cast<const char *>(g) + off),
                        reinterpret_cast<float *>(::hipsim::dyn_lds() + lds_addr));
}
inline unsigned hf_lds_addr(const void *p) { return (unsigned)(static_cast<const unsigned char *>(p) - ::hipsim::dyn_lds()); }
inline void hf_glds16_raw_s(const void *g, unsigned off, unsigned lds_addr) {
  ::hipsim::glds16(reinterpret_cast<const float *>(static_cast<const char *>(g) + off),
                   reinterpret_cast<float *>(::hipsim::dyn_lds() + lds_addr));
}

#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(m, n, s) ((void)0)
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) ::hipsim::mfma32x32x2((a), (b), (c))
typedef _Float16 hipsim_half8 __attribute__((ext_vector_type(8)));
inline ::hipsim::f32x16 hipsim_mfma_h(hipsim_half8 a, hipsim_half8 b, ::hipsim::f32x16 c) {
  float fa[8], fb[8];
  for (int k = 0; k < 8; ++k) { fa[k] = (float)a[k]; fb[k] = (float)b[k]; }
  return ::hipsim::mfma32x32x16h(fa, fb, c);
}
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) hipsim_mfma_h((a), (b), (c))

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  ::hipsim::launch([=]() { (kernel)(__VA_ARGS__); }, (grid), (block), (shmem))
