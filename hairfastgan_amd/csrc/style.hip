// style.hip - per-layer style plumbing of ModulatedConv2d.
//
// hf_modconv_prepare_f32 : one-time weight re-layout (frozen params, model.py:223-225)
// hf_modulation_f32      : EqualLinear of the modulation (model.py:153-163, :241)
// hf_demod_f32           : demodulation coefficients (model.py:244-246)
//
// All three are tiny GEMV-shaped, weight-bandwidth bound problems
// (<= 1 MB of weights per call): one wave per output row, 16 B per lane loads,
// cross-lane sum by xor shuffles.
#include "hf_common.h"

// Floating-point contraction: "on" = a multiply and an add are fused only where they are written in ONE expression (or as
// fmaf), never across statements.  hipcc's default ("fast") lets the backend fuse opportunistically per basic block: the
// unrolled body of a grid-stride loop and its remainder iterations then round differently, i.e. a sample's bits depend on
// how many elements the launch has - on what it is batched with (found by tools/probes/batch_variance.py in
// upsample_bilinear_add: the third of three images differed from the third of six by one ulp).
#pragma clang fp contract(on)

namespace {

// wt[tap][ci][co] = scale*w[co][ci][tap];  wsq[co][ci] = sum_tap (scale*w)^2
__global__ __launch_bounds__(256) void prepare_weights(float *__restrict__ wt, float *__restrict__ wsq,
                                                       const float *__restrict__ weight, int cout, int cin,
                                                       int taps, float scale) {
  long long n = (long long)cout * cin;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    int ci = (int)(i % cin);
    int co = (int)(i / cin);
    const float *src = weight + i * taps;
    float sq = 0.0f;
    for (int t = 0; t < taps; ++t) {
      float v = src[t] * scale;
      wt[((long long)t * cin + ci) * cout + co] = v;
      sq = fmaf(v, v, sq);
    }
    if (wsq) wsq[i] = sq;
  }
}

// One wave per (ci, b): both operand rows are loaded with independent 4-byte loads
// (8 per lane for style_dim 512) before the shuffle reduction, so a call costs one
// memory round trip instead of one per batch element.
constexpr int kMaxPerLane = 16;

__global__ __launch_bounds__(256) void modulation_kernel(float *__restrict__ s,
                                                         const float *__restrict__ latent,
                                                         long long lat_stride,
                                                         const float *__restrict__ mod_w,
                                                         const float *__restrict__ mod_b, int batch, int cin,
                                                         int style_dim, float scale) {
  const int lane = threadIdx.x & 63;
  const int ci = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.y;
  if (ci >= cin) return;
  const float *wr = mod_w + (long long)ci * style_dim;
  const float *lat = latent + (long long)b * lat_stride;
  float wv[kMaxPerLane], lv[kMaxPerLane];
#pragma unroll
  for (int k = 0; k < kMaxPerLane; ++k) {
    const int j = k * 64 + lane;
    const bool ok = j < style_dim;
    wv[k] = ok ? wr[j] : 0.0f;
    lv[k] = ok ? lat[j] : 0.0f;
  }
  float acc = 0.0f;
#pragma unroll
  for (int k = 0; k < kMaxPerLane; ++k) acc = fmaf(lv[k], wv[k] * scale, acc);
  acc = hf_wave_sum(acc);
  if (lane == 0) s[(long long)b * cin + ci] = acc + mod_b[ci];
}

// One wave per (b, co): d = rsqrt(sum_ci wsq[co,ci]*s[b,ci]^2 + eps)
__global__ __launch_bounds__(256) void demod_kernel(float *__restrict__ d, const float *__restrict__ s,
                                                    const float *__restrict__ wsq, int batch, int cin,
                                                    int cout) {
  const int lane = threadIdx.x & 63;
  const int co = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.y;
  if (co >= cout) return;
  const float *wr = wsq + (long long)co * cin;
  const float *sr = s + (long long)b * cin;
  float acc = 0.0f;
  for (int j = lane; j < cin; j += 64) {
    float sv = sr[j];
    acc = fmaf(wr[j], sv * sv, acc);
  }
  acc = hf_wave_sum(acc);
  if (lane == 0) d[(long long)b * cout + co] = rsqrtf(acc + 1e-8f);
}

// ---- all layers of a forward in two launches (hf_style_batch_f32) ----
// blockIdx.z = job; same arithmetic per (job, b, channel) as modulation_kernel / demod_kernel.  Round 6: a wave keeps its
// weight row for kStyleBC batch elements (blockIdx.y = batch chunk) instead of one - at batch 8 the 27 MB of modulation
// weights of a forward were streamed eight times (27 + 25 us for two launches of GEMV rows); every (b, channel) result is
// the same chain of operations as before: equal bits at any batch size.
constexpr int kStyleBC = 8;

// KN = style-vector elements per lane (8: style_dim <= 512, the generator's; 16: up to 1024); a wave computes kStyleCI
// modulation channels for kStyleBC batch elements from ONE copy of the latent rows (the first form of this round - one channel
// per wave, 16 predicated elements per lane - needed 158 registers and ran no faster than before: 13 k waves of three
// dependent round trips at three waves per SIMD)
constexpr int kStyleCI = 4;
template <int KN>
__global__ __launch_bounds__(256) void modulation_batch_kernel(float *__restrict__ out,
                                                               const float *__restrict__ latent,
                                                               long long lat_bstride, long long lat_rstride,
                                                               const hf_style_job *__restrict__ jobs, int style_dim,
                                                               float scale, int batch) {
  const hf_style_job J = jobs[blockIdx.z];
  const int lane = threadIdx.x & 63;
  const int ci0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * kStyleCI;
  const int b0 = blockIdx.y * kStyleBC;
  if (ci0 >= J.cin) return;
  const float *lat0 = latent + (long long)J.style_row * lat_rstride;
  float ws[kStyleCI][KN], lv[kStyleBC][KN];
#pragma unroll
  for (int c = 0; c < kStyleCI; ++c) {
    const float *wr = J.mod_w + (long long)min(ci0 + c, J.cin - 1) * style_dim;  // (surplus rows re-read the last one)
#pragma unroll
    for (int k = 0; k < KN; ++k) {
      const int j = k * 64 + lane;
      ws[c][k] = j < style_dim ? wr[j] : 0.0f;
    }
  }
#pragma unroll
  for (int bb = 0; bb < kStyleBC; ++bb) {
    const float *lat = lat0 + (long long)min(b0 + bb, batch - 1) * lat_bstride;
#pragma unroll
    for (int k = 0; k < KN; ++k) {
      const int j = k * 64 + lane;
      lv[bb][k] = j < style_dim ? lat[j] : 0.0f;
    }
  }
#pragma unroll
  for (int c = 0; c < kStyleCI; ++c) {
#pragma unroll
    for (int k = 0; k < KN; ++k) ws[c][k] = ws[c][k] * scale;
    const bool cok = ci0 + c < J.cin;
    const float bias = cok ? J.mod_b[ci0 + c] : 0.0f;
#pragma unroll
    for (int bb = 0; bb < kStyleBC; ++bb) {
      float acc = 0.0f;
#pragma unroll
      for (int k = 0; k < KN; ++k) acc = fmaf(lv[bb][k], ws[c][k], acc);
      acc = hf_wave_sum(acc);
      if (lane == 0 && cok && b0 + bb < batch) out[J.s_ofs + (long long)(b0 + bb) * J.cin + ci0 + c] = acc + bias;
    }
  }
}

__global__ __launch_bounds__(256) void demod_batch_kernel(float *__restrict__ out,
                                                          const hf_style_job *__restrict__ jobs, int batch) {
  const hf_style_job J = jobs[blockIdx.z];
  const int lane = threadIdx.x & 63;
  const int co = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int b0 = blockIdx.y * kStyleBC;
  if (!J.wsq || co >= J.cout) return;
  const float *wr = J.wsq + (long long)co * J.cin;
  const float *sr[kStyleBC];
  float acc[kStyleBC];
#pragma unroll
  for (int bb = 0; bb < kStyleBC; ++bb) {
    sr[bb] = out + J.s_ofs + (long long)min(b0 + bb, batch - 1) * J.cin;
    acc[bb] = 0.0f;
  }
  for (int j = lane; j < J.cin; j += 64) {
    const float wv = wr[j];
#pragma unroll
    for (int bb = 0; bb < kStyleBC; ++bb) {
      const float sv = sr[bb][j];
      acc[bb] = fmaf(wv, sv * sv, acc[bb]);
    }
  }
#pragma unroll
  for (int bb = 0; bb < kStyleBC; ++bb) {
    const float a = hf_wave_sum(acc[bb]);
    if (lane == 0 && b0 + bb < batch) out[J.d_ofs + (long long)(b0 + bb) * J.cout + co] = rsqrtf(a + 1e-8f);
  }
}

// ---- range normalisation of a (modulation, demodulation) pair (hf_style_normalize_f32) ----
// One block per (job, b): e = floor(log2 max_ci |s[b,ci]|), then s[b,:] *= 2^-e and d[b,:] *= 2^e,
// so that max|s| lies in [1, 2): |s*x| <= 2|x|, the split then overflows only where x itself would.  The demodulated conv is invariant under this rescaling
// (y = d * sum w (s x)), and a power of two makes it exact in fp32 - every fp32 consumer computes
// bit-identical results - while the fp16 (hi, lo) split of s*x in the matrix-core modes then
// depends on the activation's magnitude only: it cannot overflow through a large trained style,
// and small styles do not push s*x into the fp16 subnormals (hf_common.h, hf_split_f16).
// (StyleGAN2-ADA's fp16 layers pre-normalise styles by their infinity norm for the same reason.)
__device__ __forceinline__ void normalize_pair(float *__restrict__ s, float *__restrict__ d, int cin, int cout) {
  HF_DYN_LDS;
  float *red = reinterpret_cast<float *>(hf_dyn_lds);  // [4]
  float m = 0.0f;
  for (int i = threadIdx.x; i < cin; i += blockDim.x) m = fmaxf(m, fabsf(s[i]));
#pragma unroll
  for (int k = 32; k >= 1; k >>= 1) m = fmaxf(m, __shfl_xor(m, k, HF_WAVE));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const int e = (int)((__float_as_uint(m) >> 23) & 255u) - 127;  // floor(log2 max|s|)
  if (e <= -120 || e >= 120) return;                               // zero / denormal / huge: leave alone
  const int k = e;
  if (k == 0) return;
  const float down = __uint_as_float((unsigned int)(127 - k) << 23), up = __uint_as_float((unsigned int)(127 + k) << 23);
  for (int i = threadIdx.x; i < cin; i += blockDim.x) s[i] *= down;
  for (int i = threadIdx.x; i < cout; i += blockDim.x) d[i] *= up;
}
__global__ __launch_bounds__(256) void normalize_batch_kernel(float *__restrict__ out, const hf_style_job *__restrict__ jobs) {
  const hf_style_job J = jobs[blockIdx.y];
  if (!J.wsq) return;  // no demodulation (ToRGB): nothing could absorb the factor
  const int b = blockIdx.x;
  normalize_pair(out + J.s_ofs + (long long)b * J.cin, out + J.d_ofs + (long long)b * J.cout, J.cin, J.cout);
}
__global__ __launch_bounds__(256) void normalize_kernel(float *__restrict__ s, float *__restrict__ d, int cin, int cout) {
  const int b = blockIdx.x;
  normalize_pair(s + (long long)b * cin, d + (long long)b * cout, cin, cout);
}

}  // namespace

extern "C" int hf_style_normalize_f32(float *s, float *d, int batch, int cin, int cout, void *stream) {
  if (!s || !d || batch <= 0 || batch > 65535 || cin <= 0 || cout <= 0) return HF_E_INVALID;
  hipLaunchKernelGGL(normalize_kernel, dim3(batch), dim3(256), 16, (hipStream_t)stream, s, d, cin, cout);
  return hf_launch_status();
}

extern "C" int hf_style_batch_f32(float *out, const float *latent, long long lat_bstride, long long lat_rstride,
                                  const hf_style_job *jobs, int n_jobs, int batch, int style_dim, int max_cin,
                                  int max_cout, void *stream) {
  if (!out || !latent || !jobs || n_jobs <= 0 || n_jobs > 65535 || batch <= 0 || batch > 65535 || style_dim <= 0 ||
      style_dim > 64 * kMaxPerLane || max_cin <= 0)
    return HF_E_INVALID;
  const float scale = 1.0f / sqrtf((float)style_dim);
  const dim3 mgrid(hf_cdiv(max_cin, 4 * kStyleCI), hf_cdiv(batch, kStyleBC), n_jobs);
  if (style_dim <= 512)
    hipLaunchKernelGGL(modulation_batch_kernel<8>, mgrid, dim3(256), 0, (hipStream_t)stream, out, latent, lat_bstride, lat_rstride,
                       jobs, style_dim, scale, batch);
  else
    hipLaunchKernelGGL(modulation_batch_kernel<kMaxPerLane>, mgrid, dim3(256), 0, (hipStream_t)stream, out, latent, lat_bstride,
                       lat_rstride, jobs, style_dim, scale, batch);
  if (max_cout > 0)
  {
    hipLaunchKernelGGL(demod_batch_kernel, dim3(hf_cdiv(max_cout, 4), hf_cdiv(batch, kStyleBC), n_jobs), dim3(256), 0,
                       (hipStream_t)stream, out, jobs, batch);
    hipLaunchKernelGGL(normalize_batch_kernel, dim3(batch, n_jobs), dim3(256), 16, (hipStream_t)stream, out, jobs);
  }
  return hf_launch_status();
}

extern "C" int hf_modconv_prepare_f32(float *wt, float *wsq, const float *weight, int cout, int cin, int k,
                                      void *stream) {
  if (!wt || !weight || cout <= 0 || cin <= 0 || (k != 1 && k != 3)) return HF_E_INVALID;
  const float scale = 1.0f / sqrtf((float)(cin * k * k));
  long long n = (long long)cout * cin;
  long long g = (n + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(prepare_weights, dim3((int)g), dim3(256), 0, (hipStream_t)stream, wt, wsq, weight, cout,
                     cin, k * k, scale);
  return hf_launch_status();
}

extern "C" int hf_modulation_f32(float *s, const float *latent, long long lat_stride, const float *mod_w,
                                 const float *mod_b, int batch, int cin, int style_dim, void *stream) {
  if (!s || !latent || !mod_w || !mod_b || batch <= 0 || cin <= 0 || style_dim <= 0 ||
      style_dim > 64 * kMaxPerLane)
    return HF_E_INVALID;
  const float scale = 1.0f / sqrtf((float)style_dim);
  if (batch > 65535) return HF_E_INVALID;
  hipLaunchKernelGGL(modulation_kernel, dim3(hf_cdiv(cin, 4), batch), dim3(256), 0, (hipStream_t)stream, s, latent,
                     lat_stride, mod_w, mod_b, batch, cin, style_dim, scale);
  return hf_launch_status();
}

extern "C" int hf_demod_f32(float *d, const float *s, const float *wsq, int batch, int cin, int cout,
                            void *stream) {
  if (!d || !s || !wsq || batch <= 0 || cin <= 0 || cout <= 0 || batch > 65535) return HF_E_INVALID;
  hipLaunchKernelGGL(demod_kernel, dim3(hf_cdiv(cout, 4), batch), dim3(256), 0, (hipStream_t)stream, d, s,
                     wsq, batch, cin, cout);
  return hf_launch_status();
}
