// Probe: fragment layout of v_mfma_f32_32x32x16_f16 on gfx950 (A/B: 8 halves per lane).
// Hypothesis: A[i][k]: lane l holds row i = l&31, k = 8*(l>>5) + e (e = 0..7);
//             B[k][j]: lane l holds col j = l&31, k = 8*(l>>5) + e;
//             D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const float* A, const float* B, float* D) {  // A[32][16], B[16][32] row-major
  int l = threadIdx.x, i = l & 31, h = l >> 5;
  half8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)A[i * 16 + 8 * h + e]; b[e] = (_Float16)B[(8 * h + e) * 32 + i]; }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = c[r];
}
int main() {
  float hA[32 * 16], hB[16 * 32], hD[32 * 32], ref[32 * 32];
  for (int i = 0; i < 32; ++i) for (int kk = 0; kk < 16; ++kk) hA[i * 16 + kk] = (float)((i * 7 + kk * 3) % 11 - 5) * 0.25f;
  for (int kk = 0; kk < 16; ++kk) for (int j = 0; j < 32; ++j) hB[kk * 32 + j] = (float)((kk * 5 + j * 2 + (j > 16)) % 13 - 6) * 0.5f;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int kk = 0; kk < 16; ++kk) s += hA[i * 16 + kk] * hB[kk * 32 + j]; ref[i * 32 + j] = s; }
  float *dA, *dB, *dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  double err = 0; for (int i = 0; i < 1024; ++i) err = fmax(err, fabs(hD[i] - ref[i]));
  printf("mfma_f32_32x32x16_f16 layout probe: max err %g (%s)\n", err, err < 1e-3 ? "LAYOUT OK" : "LAYOUT MISMATCH");
  return 0;
}
