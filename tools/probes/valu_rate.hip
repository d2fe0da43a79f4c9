// Micro-benchmark (kernel development, not product): issue cost of the VALU instructions and output-store shapes of the fused
// upsampling epilogue (csrc/convh.hip, epilogue_fused), per wave-instruction, with one and two waves per SIMD on every CU.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate tools/probes/valu_rate.hip && ./valu_rate
// Each VALU test: 8 independent register chains, 16 instructions per loop body (2 per chain), s_memtime around the loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

enum { T_FMA, T_CVT_F16, T_CVT_F32, T_CVT_PK, T_DPP_MOV, T_FMAC_DPP, T_MAX, T_MED3, T_PK_FMA, T_PK_MUL, T_SWAP, T_SPLIT, T_N };
static const char *names[T_N] = {"v_fma_f32", "v_cvt_f16_f32", "v_cvt_f32_f16", "v_cvt_pk_f16_f32", "v_mov_b32_dpp wave_shr", "v_fmac_f32_dpp wave_shr",
                                 "v_max_f32", "v_med3_f32", "v_pk_fma_f32 (2 flops-pairs)", "v_pk_mul_f32", "v_permlane32_swap_b32",
                                 "split chain cvt,cvt,sub,cvt (4 instr)"};

template <int T>
__global__ __launch_bounds__(512, 2) void kv(float *out, int iters, unsigned long long *cyc) {
  float r[8], s[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = 1.0f + 0.001f * (threadIdx.x + i), s[i] = 0.5f + 0.002f * (threadIdx.x + i);
  float k = 0.999f + 1e-6f * threadIdx.x;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p[8], kk;
  kk.x = k; kk.y = k;
#pragma unroll
  for (int i = 0; i < 8; ++i) p[i].x = r[i], p[i].y = s[i];
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (T == T_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(k), "v"(s[i]));
        if (T == T_CVT_F16) asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(r[i]) : "v"(s[i]));
        if (T == T_CVT_F32) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(r[i]) : "v"(s[i]));
        if (T == T_CVT_PK) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r[i]) : "v"(s[i]), "v"(k));
        if (T == T_DPP_MOV) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r[i]) : "v"(s[i]));
        if (T == T_FMAC_DPP) asm volatile("v_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r[i]) : "v"(s[i]), "v"(k));
        if (T == T_MAX) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r[i]) : "v"(s[i]));
        if (T == T_MED3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(s[i]), "v"(k));
        if (T == T_PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(kk), "v"(p[(i + 1) & 7]));
        if (T == T_PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(kk));
        if (T == T_SWAP) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(r[i]), "+v"(s[i]));
      }
    }
    if (T == T_SPLIT) {  // the hi / lo split of 8 values as the compiler schedules it (32 instructions, counted as 16 below -> x2)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float hi, back, lo;
        asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(hi) : "v"(r[i]));
        asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(back) : "v"(hi));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(lo) : "v"(r[i]), "v"(back));
        asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(s[i]) : "v"(lo));
        r[i] = s[i] + hi;
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float sum = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i) sum += r[i] + s[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// Store shapes: every wave of the block issues bursts of 16 stores (as one channel quad of the epilogue), then waits (vmcnt 0).
//   S=0: dwordx2, lane stride 32 B + 8 B for the upper half-wave (today: half of every 16-byte unit pair per instruction)
//   S=1: dwordx4, lanes 0-31 at 32-byte stride, lanes 32-63 in a second array (round 4's HF_H_SPLIT_STORE16)
//   S=2: dwordx4, 64 lanes contiguous (1 KiB per instruction)
//   S=3: dwordx2, 64 lanes contiguous (512 B per instruction)
template <int S>
__global__ __launch_bounds__(512, 2) void ks(char *buf, int iters, unsigned long long *cyc, long long wave_span) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, lh = lane >> 5;
  char *base = buf + ((long long)blockIdx.x * 8 + wave) * wave_span;
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  u2 v2;
  v2.x = lane; v2.y = wave;
  u4 v4;
  v4.x = lane; v4.y = wave; v4.z = lane; v4.w = wave;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    char *p = base + (long long)(it & 15) * 32768;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (S == 0) *reinterpret_cast<u2 *>(p + (j >> 1) * 2048 + li * 32 + (j & 1) * 16 + lh * 8) = v2;
      if (S == 1 && j < 8) *reinterpret_cast<u4 *>(p + lh * 16384 + j * 2048 + li * 32) = v4;
      if (S == 2 && j < 8) *reinterpret_cast<u4 *>(p + j * 1024 + lane * 16) = v4;
      if (S == 3) *reinterpret_cast<u2 *>(p + j * 512 + lane * 8) = v2;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int T>
void runv(int threads, int iters, float *out, unsigned long long *cyc) {
  const int blocks = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(kv<T>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kv<T>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h;
  hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const double n = (double)iters * (T == T_SPLIT ? 40 : 16);
  printf("%-44s %d wave(s)/SIMD: %6.2f memtime ticks per wave-instruction (block 0), %7.3f ms, %5.2f ns*CU-SIMD per instr\n", names[T], threads / 256,
         (double)h / n, ms, ms * 1e6 / (n * (threads / 256)));
}

template <int S>
void runs(const char *name, int threads, int iters, char *buf, unsigned long long *cyc) {
  const int blocks = 256;
  const long long span = 16LL * 32768;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(ks<S>, dim3(blocks), dim3(threads), 0, 0, buf, iters, cyc, span);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(ks<S>, dim3(blocks), dim3(threads), 0, 0, buf, iters, cyc, span);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h;
  hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const double bytes_per_burst = 8192.0;  // per wave: 16 x 512 B or 8 x 1 KiB
  const double total = bytes_per_burst * (threads / 64) * blocks * iters;
  printf("%-58s %d waves/CU: %8.1f ticks per 8 KiB burst per wave (block 0), %7.3f ms, %6.2f TB/s, %5.2f B/clk/CU at 2.4 GHz\n", name, threads / 64,
         (double)h / iters, ms, total / ms * 1e-9, total / ms * 1e-9 * 1e12 / 256 / 2.4e9);
}

int main() {
  float *out;
  unsigned long long *cyc;
  char *buf;
  hipMalloc(&out, sizeof(float) * 256 * 512);
  hipMalloc(&cyc, sizeof(unsigned long long) * 256);
  hipMalloc(&buf, 256LL * 8 * 16 * 32768);
  for (int threads : {256, 512}) {
    const int it = 20000;
    runv<T_FMA>(threads, it, out, cyc);
    runv<T_CVT_F16>(threads, it, out, cyc);
    runv<T_CVT_F32>(threads, it, out, cyc);
    runv<T_CVT_PK>(threads, it, out, cyc);
    runv<T_DPP_MOV>(threads, it, out, cyc);
    runv<T_FMAC_DPP>(threads, it, out, cyc);
    runv<T_MAX>(threads, it, out, cyc);
    runv<T_MED3>(threads, it, out, cyc);
    runv<T_PK_FMA>(threads, it, out, cyc);
    runv<T_PK_MUL>(threads, it, out, cyc);
    runv<T_SWAP>(threads, it, out, cyc);
    runv<T_SPLIT>(threads, it, out, cyc);
  }
  for (int threads : {256, 512}) {
    const int it = 2000;
    runs<0>("dwordx2, 32-B lane stride + half-wave 8 B (today)", threads, it, buf, cyc);
    runs<1>("dwordx4, 32-B lane stride, half-waves in two arrays", threads, it, buf, cyc);
    runs<2>("dwordx4, 64 lanes contiguous", threads, it, buf, cyc);
    runs<3>("dwordx2, 64 lanes contiguous", threads, it, buf, cyc);
  }
  return 0;
}
