// elementwise.hip - streaming bias / noise / leaky-ReLU kernels (HBM bound).
//
// hf_fused_bias_act_f32   : FusedLeakyReLU forward  (reference op/fused_act.py:85-96,
//                           fused_bias_act_kernel.cu:18-49 case act*10+grad == 30)
// hf_noise_bias_act_f32   : NoiseInjection + FusedLeakyReLU (model.py:288-293, :341)
//
// Algorithmic traffic: 8 B per element (one fp32 read, one write); noise/bias
// are tiny and L2 resident.  16 B per lane accesses, grid-stride, <= 2048 blocks.
#include "hf_common.h"

// Floating-point contraction: "on" = a multiply and an add are fused only where they are written in ONE expression (or as
// fmaf), never across statements.  hipcc's default ("fast") lets the backend fuse opportunistically per basic block: the
// unrolled body of a grid-stride loop and its remainder iterations then round differently, i.e. a sample's bits depend on
// how many elements the launch has - on what it is batched with (found by tools/probes/batch_variance.py in
// upsample_bilinear_add: the third of three images differed from the third of six by one ulp).
#pragma clang fp contract(on)

namespace {

constexpr int kThreads = 256;

// Generic form: bias index = (i / step_b) % n_bias, any alignment.
__global__ __launch_bounds__(kThreads) void bias_act_scalar(float *__restrict__ out,
                                                            const float *__restrict__ x,
                                                            const float *__restrict__ bias, long long n,
                                                            int n_bias, int step_b, float alpha,
                                                            float scale) {
  long long stride = (long long)gridDim.x * kThreads;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
    float v = x[i];
    if (bias) v += bias[(i / step_b) % n_bias];
    out[i] = hf_lrelu(v, alpha, scale);
  }
}

// Plane form: element (plane, p) with plane = b*C + c; hw % 4 == 0 so a float4
// never straddles a plane.  noise may be null.
__global__ __launch_bounds__(kThreads) void noise_bias_act_vec4(
    float *__restrict__ out, const float *__restrict__ x, const float *__restrict__ noise,
    const float *__restrict__ noise_w, const float *__restrict__ bias, int channels, int hw4,
    long long noise_bstride, long long n4, float alpha, float scale) {
  const float nw = (noise != nullptr) ? noise_w[0] : 0.0f;
  long long stride = (long long)gridDim.x * kThreads;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n4; i += stride) {
    long long plane = i / hw4;
    int p4 = (int)(i - plane * hw4);
    int c = (int)(plane % channels);
    long long b = plane / channels;
    float4 v = reinterpret_cast<const float4 *>(x)[i];
    float bc = bias ? bias[c] : 0.0f;
    if (noise) {
      float4 z = *reinterpret_cast<const float4 *>(noise + b * noise_bstride + 4LL * p4);
      v.x = fmaf(nw, z.x, v.x);
      v.y = fmaf(nw, z.y, v.y);
      v.z = fmaf(nw, z.z, v.z);
      v.w = fmaf(nw, z.w, v.w);
    }
    v.x = hf_lrelu(v.x + bc, alpha, scale);
    v.y = hf_lrelu(v.y + bc, alpha, scale);
    v.z = hf_lrelu(v.z + bc, alpha, scale);
    v.w = hf_lrelu(v.w + bc, alpha, scale);
    reinterpret_cast<float4 *>(out)[i] = v;
  }
}

__global__ __launch_bounds__(kThreads) void noise_bias_act_scalar(
    float *__restrict__ out, const float *__restrict__ x, const float *__restrict__ noise,
    const float *__restrict__ noise_w, const float *__restrict__ bias, int channels, int hw,
    long long noise_bstride, long long n, float alpha, float scale) {
  const float nw = (noise != nullptr) ? noise_w[0] : 0.0f;
  long long stride = (long long)gridDim.x * kThreads;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
    long long plane = i / hw;
    int p = (int)(i - plane * hw);
    int c = (int)(plane % channels);
    long long b = plane / channels;
    float v = x[i];
    if (noise) v = fmaf(nw, noise[b * noise_bstride + p], v);
    if (bias) v += bias[c];
    out[i] = hf_lrelu(v, alpha, scale);
  }
}

inline int stream_grid(long long work_items) {
  long long g = (work_items + kThreads - 1) / kThreads;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" int hf_fused_bias_act_f32(float *out, const float *x, const float *bias, long long n,
                                     int n_bias, int step_b, float alpha, float scale, void *stream) {
  if (!out || !x || n < 0 || (bias && (n_bias <= 0 || step_b <= 0))) return HF_E_INVALID;
  if (n == 0) return HF_OK;
  hipStream_t st = (hipStream_t)stream;
  const bool aligned = ((((size_t)out) | ((size_t)x)) & 15) == 0;
  if (bias && step_b % 4 == 0 && aligned && n % step_b == 0) {
    // planes of step_b elements: reuse the vector kernel (channels = n_bias)
    long long n4 = n / 4;
    hipLaunchKernelGGL(noise_bias_act_vec4, dim3(stream_grid(n4)), dim3(kThreads), 0, st, out, x,
                       (const float *)nullptr, (const float *)nullptr, bias, n_bias, step_b / 4, 0LL, n4,
                       alpha, scale);
  } else {
    hipLaunchKernelGGL(bias_act_scalar, dim3(stream_grid(n)), dim3(kThreads), 0, st, out, x, bias, n,
                       n_bias, step_b, alpha, scale);
  }
  return hf_launch_status();
}

extern "C" int hf_noise_bias_act_f32(float *out, const float *x, const float *noise, const float *noise_w,
                                     const float *bias, int batch, int channels, int hw,
                                     long long noise_bstride, float alpha, float scale, void *stream) {
  if (!out || !x || batch <= 0 || channels <= 0 || hw <= 0 || (noise && !noise_w)) return HF_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  long long n = (long long)batch * channels * hw;
  const bool aligned = ((((size_t)out) | ((size_t)x) | ((size_t)noise)) & 15) == 0;
  if (hw % 4 == 0 && noise_bstride % 4 == 0 && aligned) {
    long long n4 = n / 4;
    hipLaunchKernelGGL(noise_bias_act_vec4, dim3(stream_grid(n4)), dim3(kThreads), 0, st, out, x, noise,
                       noise_w, bias, channels, hw / 4, noise_bstride, n4, alpha, scale);
  } else {
    hipLaunchKernelGGL(noise_bias_act_scalar, dim3(stream_grid(n)), dim3(kThreads), 0, st, out, x, noise,
                       noise_w, bias, channels, hw, noise_bstride, n, alpha, scale);
  }
  return hf_launch_status();
}
