// Micro-benchmark (kernel development, not product): issue rate of v_mfma_f32_32x32x16_f16 in the access pattern of the
// f16x3 conv kernels - 2 x 2 accumulator tiles, three MFMAs per tile and tap (hi*hi, hi*lo, lo*hi), fragments from LDS by
// ds_read_b128 - as a function of where the accumulators live (VGPR / AGPR), of the LDS fragment reads and of the waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_rate tools/probes/mfma_rate.hip && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0)
#define MFMA_A(acc, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))

// MODE 0: MFMAs only (fragments constant in registers), acc where the compiler puts them
// MODE 1: MFMAs only, accumulators forced into AGPRs
// MODE 2: + 8 ds_read_b128 per 12 MFMAs (double-buffered fragment slots, as the conv kernels), compiler's acc
// MODE 3: as 2, accumulators in AGPRs
template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float *out, int iters, unsigned long long *cyc, int zero) {
  extern __shared__ half8 lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) {
    half8 v;
    for (int k_ = 0; k_ < 8; ++k_) v[k_] = zero ? (_Float16)0.0f : (_Float16)(0.001f * ((i * 8 + k_) % 97));
    lds[i] = v;
  }
  __syncthreads();
  f32x16 acc[2][2];
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b)
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
  half8 ah[2][2], al[2][2], bh[2][2], bl[2][2];
  const half8 *base = lds + lane + wave * 64;
  for (int s = 0; s < 2; ++s)
    for (int t = 0; t < 2; ++t) {
      ah[s][t] = base[(s * 2 + t) * 64 % 2048];
      al[s][t] = base[(s * 2 + t + 4) * 64 % 2048];
      bh[s][t] = base[(s * 2 + t + 8) * 64 % 2048];
      bl[s][t] = base[(s * 2 + t + 12) * 64 % 2048];
    }
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int s = tap & 1;
      if (MODE >= 2) {  // fragments of the next tap
        const int o = ((it * 9 + tap) * 8 * 64) & 2047;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          ah[s ^ 1][t] = base[(o + t * 64) & 2047];
          al[s ^ 1][t] = base[(o + (t + 2) * 64) & 2047];
          bh[s ^ 1][t] = base[(o + (t + 4) * 64) & 2047];
          bl[s ^ 1][t] = base[(o + (t + 6) * 64) & 2047];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (MODE & 1) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) MFMA_A(acc[a][b], ah[s][a], bh[s][b]);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) MFMA_A(acc[a][b], ah[s][a], bl[s][b]);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) MFMA_A(acc[a][b], al[s][a], bh[s][b]);
      } else {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) MFMA(acc[a][b], ah[s][a], bh[s][b]);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) MFMA(acc[a][b], ah[s][a], bl[s][b]);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) MFMA(acc[a][b], al[s][a], bh[s][b]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float sum = 0.0f;
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b)
      for (int r = 0; r < 16; ++r) sum += acc[a][b][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

static int g_zero = 0;
template <int MODE>
void run(const char *name, int threads, int iters) {
  float *out;
  unsigned long long *cyc;
  const int blocks = 256;
  hipMalloc(&out, sizeof(float) * blocks * 512);
  hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 64 * 1024, 0, out, iters, cyc, g_zero);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 64 * 1024, 0, out, iters, cyc, g_zero);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
  const double mfma_per_wave = (double)iters * 9 * 12;
  const int waves_per_simd = threads / 256;
  const double flops = mfma_per_wave * (threads / 64) * blocks * 32.0 * 32 * 16 * 2;
  printf("%-44s %d wave(s)/SIMD: %8.3f ms  %7.1f TFLOP/s (%5.1f %% of 2516.6)  block 0: %6.1f memtime ticks per MFMA per SIMD\n", name,
         waves_per_simd, ms, flops / ms * 1e-9, flops / ms * 1e-9 / 25.166, (double)h[0] / (mfma_per_wave * waves_per_simd));
  hipFree(out);
  hipFree(cyc);
}

// usage: mfma_rate [iters = 2000] [zero = 0]   zero = 1: all-zero operands (MI355X_MICROARCH.md, DVFS give-back: the chip clocks higher
// on zero data); iters sets the loop's duration (2000 = 4-8 ms; 50 = 0.1-0.2 ms, before the power management reacts; 200000 = 0.4-0.8 s)
int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  g_zero = argc > 2 ? atoi(argv[2]) : 0;
  printf("iters %d, %s operands\n", iters, g_zero ? "ZERO" : "non-zero");
  for (int threads : {512, 256}) {
    run<0>("MFMA only, acc by the compiler", threads, iters);
    run<1>("MFMA only, acc in AGPRs", threads, iters);
    run<2>("+ 8 ds_read_b128 / 12 MFMAs, compiler acc", threads, iters);
    run<3>("+ 8 ds_read_b128 / 12 MFMAs, acc in AGPRs", threads, iters);
  }
  return 0;
}
