// encoder_ops.hip - the small operators around the convolutions of the two encoders
// (e4e: models/encoder4editing/models/encoders/{psp_encoders,helpers}.py;
//  FeatureStyle: models/FeatureStyleEncoder/nets/feature_style_encoder.py, arcface/iresnet.py,
//  trainer.py:61-64).  All are streaming / latency-bound kernels on tensors of at most a
//  few MB; the FLOPs of the encoders live in modconv.hip (hf_conv2d_f32).
//
//   hf_conv_prepare_f32          Conv2d weight [cout,cin,k,k] -> wt[tap][ci][co]     (once per checkpoint)
//   hf_bn_fold_f32               BatchNorm2d (inference) -> per-channel scale/shift  (once per checkpoint)
//   hf_plane_mean_f32            AdaptiveAvgPool2d(1) of SEModule (helpers.py:60,68)
//   hf_se_gate_f32               fc1 -> ReLU -> fc2 -> sigmoid of SEModule (helpers.py:69-73)
//   hf_scale_shortcut_add_f32    res * gate + shortcut, shortcut optionally MaxPool2d(1, stride)
//                                (helpers.py:95-96, :118-120)
//   hf_upsample_bilinear_add_f32 _upsample_add (helpers.py:123-140; align_corners=True)
//   hf_adaptive_avgpool_f32      AdaptiveAvgPool2d((3,3)) + channel concat (feature_style_encoder.py:44, 52-61)
//   hf_downscale2x_f32           F.interpolate(scale_factor=0.5, 'bilinear') (trainer.py:61-64)
//   hf_linear_f32                nn.Linear / EqualLinear heads (weight-bandwidth bound GEMV batch)
//   hf_add_bcast_f32             w0 + delta_i, + latent_avg (psp_encoders.py:199, model_utils.py:9-13)
#include <cstdint>

#include "hf_common.h"

// Floating-point contraction: "on" = a multiply and an add are fused only where they are written in ONE expression (or as
// fmaf), never across statements.  hipcc's default ("fast") lets the backend fuse opportunistically per basic block: the
// unrolled body of a grid-stride loop and its remainder iterations then round differently, i.e. a sample's bits depend on
// how many elements the launch has - on what it is batched with (found by tools/probes/batch_variance.py in
// upsample_bilinear_add: the third of three images differed from the third of six by one ulp).
#pragma clang fp contract(on)

namespace {

__global__ __launch_bounds__(256) void conv_prepare(float *__restrict__ wt, const float *__restrict__ weight,
                                                    int cout, int cin, int taps, float scale) {
  const long long n = (long long)cout * cin;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int ci = (int)(i % cin);
    const int co = (int)(i / cin);
    for (int t = 0; t < taps; ++t) wt[((long long)t * cin + ci) * cout + co] = weight[i * taps + t] * scale;
  }
}

// scale = gamma / sqrt(var + eps); shift = beta + (conv_bias - mean) * scale
__global__ void bn_fold(float *__restrict__ scale, float *__restrict__ shift, const float *__restrict__ gamma,
                        const float *__restrict__ beta, const float *__restrict__ mean,
                        const float *__restrict__ var, const float *__restrict__ conv_bias, float eps, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float g = gamma[i] / sqrtf(var[i] + eps);
  scale[i] = g;
  shift[i] = beta[i] + ((conv_bias ? conv_bias[i] : 0.0f) - mean[i]) * g;
}

// one wave per plane (small planes); 16-byte loads when the plane size allows
__global__ __launch_bounds__(256) void plane_mean(float *__restrict__ out, const float *__restrict__ x,
                                                  int planes, int hw) {
  const int lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= planes) return;
  const float *src = x + (long long)p * hw;
  float acc = 0.0f;
  if ((hw & 3) == 0) {
    const float4 *s4 = reinterpret_cast<const float4 *>(src);
    for (int i = lane; i < (hw >> 2); i += 64) {
      const float4 v = s4[i];
      acc += (v.x + v.y) + (v.z + v.w);
    }
  } else {
    for (int i = lane; i < hw; i += 64) acc += src[i];
  }
  acc = hf_wave_sum(acc);
  if (lane == 0) out[p] = acc / (float)hw;
}
// one block per plane (planes of >= 4096 elements: 64 channels x 3 images are only 192 planes - a wave each left
// most of the chip idle, 15 us per call); fixed summation order
__global__ __launch_bounds__(256) void plane_mean_block(float *__restrict__ out, const float *__restrict__ x, int hw) {
  HF_DYN_LDS;
  float *part = reinterpret_cast<float *>(hf_dyn_lds);  // [4] (dynamic: the CPU interpreter shares only that)
  const float4 *s4 = reinterpret_cast<const float4 *>(x + (long long)blockIdx.x * hw);
  float acc = 0.0f;
  for (int i = threadIdx.x; i < (hw >> 2); i += 256) {
    const float4 v = s4[i];
    acc += (v.x + v.y) + (v.z + v.w);
  }
  acc = hf_wave_sum(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = ((part[0] + part[1]) + (part[2] + part[3])) / (float)hw;
}

// one block per image: hidden = relu(fc1 . pooled); gate = sigmoid(fc2 . hidden)
__global__ __launch_bounds__(256) void se_gate(float *__restrict__ gate, const float *__restrict__ pooled,
                                               const float *__restrict__ fc1, const float *__restrict__ fc2,
                                               int c, int cr) {
  HF_DYN_LDS;
  float *pl = reinterpret_cast<float *>(hf_dyn_lds);  // [c]
  float *hid = pl + c;                                 // [cr]
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < c; i += blockDim.x) pl[i] = pooled[(long long)b * c + i];
  __syncthreads();
  for (int j = threadIdx.x; j < cr; j += blockDim.x) {
    float acc = 0.0f;
    for (int i = 0; i < c; ++i) acc = fmaf(fc1[(long long)j * c + i], pl[i], acc);
    hid[j] = acc > 0.0f ? acc : 0.0f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < c; i += blockDim.x) {
    float acc = 0.0f;
    for (int j = 0; j < cr; ++j) acc = fmaf(fc2[(long long)i * cr + j], hid[j], acc);
    gate[(long long)b * c + i] = 1.0f / (1.0f + expf(-acc));
  }
}

__global__ __launch_bounds__(256) void scale_shortcut_add(float *__restrict__ out, const float *__restrict__ r,
                                                          const float *__restrict__ gate,
                                                          const float *__restrict__ shortcut, int sc_stride,
                                                          int oh, int ow, int sh, int sw, long long total) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  const int oplane = oh * ow;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long long pl = i / oplane;
    const int p = (int)(i - pl * oplane);
    const int y = p / ow, x = p - y * ow;
    float v = r[i];
    if (gate) v *= gate[pl];
    v += shortcut[pl * (long long)sh * sw + (long long)(y * sc_stride) * sw + x * sc_stride];
    out[i] = v;
  }
}

// bilinear, align_corners=True: src = dst * (in-1)/(out-1)
__global__ __launch_bounds__(256) void upsample_bilinear_add(float *__restrict__ out, const float *__restrict__ x,
                                                             const float *__restrict__ y, int h, int w, int oh,
                                                             int ow, long long total) {
  const float ry = oh > 1 ? (float)(h - 1) / (float)(oh - 1) : 0.0f;
  const float rx = ow > 1 ? (float)(w - 1) / (float)(ow - 1) : 0.0f;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const int oplane = oh * ow;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long long pl = i / oplane;
    const int p = (int)(i - pl * oplane);
    const int oy = p / ow, ox = p - oy * ow;
    const float fy = ry * oy, fx = rx * ox;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
    const float ly = fy - y0, lx = fx - x0;
    const float *src = x + pl * (long long)h * w;
    const float top = (1.0f - lx) * src[y0 * w + x0] + lx * src[y0 * w + x1];
    const float bot = (1.0f - lx) * src[y1 * w + x0] + lx * src[y1 * w + x1];
    out[i] = (1.0f - ly) * top + ly * bot + y[i];
  }
}

// one wave per (plane, output bin); bins [floor(i*H/oh), ceil((i+1)*H/oh))
__global__ __launch_bounds__(256) void adaptive_avgpool(float *__restrict__ out, const float *__restrict__ x,
                                                        int batch, int channels, int h, int w, int oh, int ow,
                                                        int ctot, int c_off) {
  const int lane = threadIdx.x & 63;
  const long long item = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long nitems = (long long)batch * channels * oh * ow;
  if (item >= nitems) return;
  const int bin = (int)(item % (oh * ow));
  const long long pl = item / (oh * ow);
  const int c = (int)(pl % channels);
  const int b = (int)(pl / channels);
  const int by = bin / ow, bx = bin - by * ow;
  const int y0 = (by * h) / oh, y1 = ((by + 1) * h + oh - 1) / oh;
  const int x0 = (bx * w) / ow, x1 = ((bx + 1) * w + ow - 1) / ow;
  const int bw = x1 - x0, n = (y1 - y0) * bw;
  const float *src = x + pl * (long long)h * w;
  float acc = 0.0f;
  for (int i = lane; i < n; i += 64) acc += src[(y0 + i / bw) * w + x0 + i % bw];
  acc = hf_wave_sum(acc);
  if (lane == 0) out[(((long long)b * ctot + c_off + c) * oh + by) * ow + bx] = acc / (float)n;
}

// scale_factor 0.5, bilinear, align_corners=False: every output is the centre of a 2x2 cell;
// same operation order as ATen's upsample_bilinear2d (rows blended, then columns).
__global__ __launch_bounds__(256) void downscale2x(float *__restrict__ out, const float *__restrict__ x, int h,
                                                   int w, long long total) {
  const int oh = h / 2, ow = w / 2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const int oplane = oh * ow;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long long pl = i / oplane;
    const int p = (int)(i - pl * oplane);
    const int oy = p / ow, ox = p - oy * ow;
    const float *src = x + pl * (long long)h * w + (long long)(2 * oy) * w + 2 * ox;
    const float2 a = *reinterpret_cast<const float2 *>(src);
    const float2 b = *reinterpret_cast<const float2 *>(src + w);
    out[i] = 0.5f * (0.5f * a.x + 0.5f * a.y) + 0.5f * (0.5f * b.x + 0.5f * b.y);
  }
}

// out[b, n] = scale * sum_k x[b*x_stride + k] * w[n*in_f + k] + bias[n]; one wave per ROWS output
// rows, the weight rows are streamed once (16 B per lane), x comes from L1/L2.
constexpr int kLinRows = 4;
constexpr int kLinMaxBatch = 8;

__global__ __launch_bounds__(256) void linear_kernel(float *__restrict__ out, const float *__restrict__ x,
                                                     long long x_stride, const float *__restrict__ w,
                                                     const float *__restrict__ bias, int batch, int in_f,
                                                     int out_f, float scale, float bias_scale, int act, float alpha,
                                                     float act_scale) {
  const int lane = threadIdx.x & 63;
  const int n0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * kLinRows;
  if (n0 >= out_f) return;
  // blockIdx.y: chunk of kLinMaxBatch input rows (any number of rows in one launch)
  x += (long long)blockIdx.y * kLinMaxBatch * x_stride;
  out += (long long)blockIdx.y * kLinMaxBatch * out_f;
  batch = min(kLinMaxBatch, batch - (int)blockIdx.y * kLinMaxBatch);
  float acc[kLinRows][kLinMaxBatch];
#pragma unroll
  for (int r = 0; r < kLinRows; ++r)
#pragma unroll
    for (int b = 0; b < kLinMaxBatch; ++b) acc[r][b] = 0.0f;
  const bool vec = (in_f & 3) == 0 && (x_stride & 3) == 0 &&
                   ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 15) == 0;
  if (vec) {
    for (int k = lane * 4; k < in_f; k += 256) {
      float4 xv[kLinMaxBatch];
#pragma unroll
      for (int b = 0; b < kLinMaxBatch; ++b)
        xv[b] = (b < batch) ? *reinterpret_cast<const float4 *>(x + b * x_stride + k) : make_float4(0, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < kLinRows; ++r) {
        const int n = min(n0 + r, out_f - 1);
        const float4 wv = *reinterpret_cast<const float4 *>(w + (long long)n * in_f + k);
#pragma unroll
        for (int b = 0; b < kLinMaxBatch; ++b)
          acc[r][b] = fmaf(wv.x, xv[b].x, fmaf(wv.y, xv[b].y, fmaf(wv.z, xv[b].z, fmaf(wv.w, xv[b].w, acc[r][b]))));
      }
    }
  } else {
    for (int k = lane; k < in_f; k += 64) {
#pragma unroll
      for (int r = 0; r < kLinRows; ++r) {
        const int n = min(n0 + r, out_f - 1);
        const float wv = w[(long long)n * in_f + k];
#pragma unroll
        for (int b = 0; b < kLinMaxBatch; ++b)
          if (b < batch) acc[r][b] = fmaf(wv, x[b * x_stride + k], acc[r][b]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < kLinRows; ++r)
#pragma unroll
    for (int b = 0; b < kLinMaxBatch; ++b) {
      const float v = hf_wave_sum(acc[r][b]);
      if (lane == 0 && b < batch && n0 + r < out_f) {
        float y = v * scale + (bias ? bias[n0 + r] * bias_scale : 0.0f);
        if (act) y = hf_lrelu(y, alpha, act_scale);
        out[(long long)b * out_f + n0 + r] = y;
      }
    }
}

// The same product for LONG rows (in_f >= 4096: the 8192- and 16384-wide Linear layers of the shape adaptor and the
// e4e / FS style heads): the four waves of a block share kLinRows output rows and split K between them, partial sums
// meet in LDS in wave order (deterministic).  One wave per row group left half of the chip idle with only 4 KiB of
// weights in flight per wave (0.85 TB/s on a 134 MB weight matrix); K-split quadruples the waves.
__global__ __launch_bounds__(256) void linear_kernel_ksplit(float *__restrict__ out, const float *__restrict__ x,
                                                            long long x_stride, const float *__restrict__ w,
                                                            const float *__restrict__ bias, int batch, int in_f, int out_f,
                                                            float scale, float bias_scale, int act, float alpha,
                                                            float act_scale) {
  HF_DYN_LDS;
  float *red = reinterpret_cast<float *>(hf_dyn_lds);  // [4 waves][kLinRows * kLinMaxBatch]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = blockIdx.x * kLinRows;
  x += (long long)blockIdx.y * kLinMaxBatch * x_stride;
  out += (long long)blockIdx.y * kLinMaxBatch * out_f;
  batch = min(kLinMaxBatch, batch - (int)blockIdx.y * kLinMaxBatch);
  float acc[kLinRows][kLinMaxBatch];
#pragma unroll
  for (int r = 0; r < kLinRows; ++r)
#pragma unroll
    for (int b = 0; b < kLinMaxBatch; ++b) acc[r][b] = 0.0f;
  const int kq = in_f / 4, k_lo = wave * kq, k_hi = wave == 3 ? in_f : k_lo + kq;  // in_f % 16 == 0 (launcher)
  for (int k = k_lo + lane * 4; k < k_hi; k += 256) {
    float4 xv[kLinMaxBatch];
#pragma unroll
    for (int b = 0; b < kLinMaxBatch; ++b)
      xv[b] = (b < batch) ? *reinterpret_cast<const float4 *>(x + b * x_stride + k) : make_float4(0, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < kLinRows; ++r) {
      const int n = min(n0 + r, out_f - 1);
      const float4 wv = *reinterpret_cast<const float4 *>(w + (long long)n * in_f + k);
#pragma unroll
      for (int b = 0; b < kLinMaxBatch; ++b)
        acc[r][b] = fmaf(wv.x, xv[b].x, fmaf(wv.y, xv[b].y, fmaf(wv.z, xv[b].z, fmaf(wv.w, xv[b].w, acc[r][b]))));
    }
  }
#pragma unroll
  for (int r = 0; r < kLinRows; ++r)
#pragma unroll
    for (int b = 0; b < kLinMaxBatch; ++b) {
      const float v = hf_wave_sum(acc[r][b]);
      if (lane == 0) red[wave * (kLinRows * kLinMaxBatch) + r * kLinMaxBatch + b] = v;
    }
  __syncthreads();
  if (threadIdx.x < kLinRows * kLinMaxBatch) {
    const int r = threadIdx.x / kLinMaxBatch, b = threadIdx.x % kLinMaxBatch;
    if (b < batch && n0 + r < out_f) {
      const int i = r * kLinMaxBatch + b, st = kLinRows * kLinMaxBatch;
      float y = ((red[i] + red[st + i]) + (red[2 * st + i] + red[3 * st + i])) * scale + (bias ? bias[n0 + r] * bias_scale : 0.0f);
      if (act) y = hf_lrelu(y, alpha, act_scale);
      out[(long long)b * out_f + n0 + r] = y;
    }
  }
}

// PixelNorm (models/stylegan2/model.py:16-21): x * rsqrt(mean(x^2, dim 1) + 1e-8); one wave per row
__global__ __launch_bounds__(256) void pixel_norm_rows(float *__restrict__ out, const float *__restrict__ x, int rows,
                                                       int dim) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float *xr = x + (long long)r * dim;
  float acc = 0.0f;
  for (int k = lane; k < dim; k += 64) acc = fmaf(xr[k], xr[k], acc);
  acc = hf_wave_sum(acc);
  const float inv = rsqrtf(acc / (float)dim + 1e-8f);
  for (int k = lane; k < dim; k += 64) out[(long long)r * dim + k] = xr[k] * inv;
}

// LayerNorm over the last `dim` elements of each row (F.layer_norm: biased variance, eps inside the
// sqrt), optional elementwise affine (gamma/beta [dim]) and LeakyReLU(alpha) on top; one block per row.
// groups > 1: gamma / beta are [groups][dim] and row r takes those of group r % groups (several LayerNorms with different
// affines over the columns of one stacked Linear output, viewed as [rows * groups, dim])
__global__ __launch_bounds__(256) void layernorm_rows(float *__restrict__ out, const float *__restrict__ x,
                                                      const float *__restrict__ gamma, const float *__restrict__ beta, int dim,
                                                      float eps, int act, float alpha, int groups) {
  HF_DYN_LDS;
  float *red = reinterpret_cast<float *>(hf_dyn_lds);  // [8]
  const long long gofs = (long long)((int)blockIdx.x % groups) * dim;
  if (gamma) gamma += gofs;
  if (beta) beta += gofs;
  const float *xr = x + (long long)blockIdx.x * dim;
  float *orow = out + (long long)blockIdx.x * dim;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float s = 0.0f;
  for (int k = threadIdx.x; k < dim; k += 256) s += xr[k];
  s = hf_wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)dim;
  float v = 0.0f;
  for (int k = threadIdx.x; k < dim; k += 256) {
    const float d = xr[k] - mean;
    v = fmaf(d, d, v);
  }
  v = hf_wave_sum(v);
  if (lane == 0) red[4 + wave] = v;
  __syncthreads();
  const float inv = rsqrtf((red[4] + red[5] + red[6] + red[7]) / (float)dim + eps);
  for (int k = threadIdx.x; k < dim; k += 256) {
    float y = (xr[k] - mean) * inv;
    if (gamma) y = fmaf(y, gamma[k], beta ? beta[k] : 0.0f);
    if (act) y = y > 0.0f ? y : y * alpha;
    orow[k] = y;
  }
}

// out = x * (1 + gamma) + beta, optionally LeakyReLU(alpha)  (ModulationModule.forward, Encoders.py:29-31)
__global__ __launch_bounds__(256) void modulate_kernel(float *__restrict__ out, const float *__restrict__ x,
                                                       const float *__restrict__ gamma, const float *__restrict__ beta, long long n,
                                                       int act, float alpha) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float y = fmaf(x[i], 1.0f + gamma[i], beta[i]);
    if (act) y = y > 0.0f ? y : y * alpha;
    out[i] = y;
  }
}

// PixelNorm over dim 1 of [B, L, D] (the reference applies models/stylegan2/model.py:16-21 to W+ codes:
// the mean runs over the L = 18 layers, Encoders.py:123-124): one thread per (b, d)
__global__ __launch_bounds__(256) void pixel_norm_dim1(float *__restrict__ out, const float *__restrict__ x, int batch, int L,
                                                       int D) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= batch * D) return;
  const int b = i / D, d = i - b * D;
  const float *xb = x + (long long)b * L * D + d;
  float acc = 0.0f;
  for (int l = 0; l < L; ++l) acc = fmaf(xb[(long long)l * D], xb[(long long)l * D], acc);
  const float inv = rsqrtf(acc / (float)L + 1e-8f);
  for (int l = 0; l < L; ++l) out[(long long)b * L * D + (long long)l * D + d] = xb[(long long)l * D] * inv;
}

__global__ __launch_bounds__(256) void add_bcast(float *__restrict__ out, const float *__restrict__ a,
                                                 const float *__restrict__ b, long long n, long long period) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = a[i] + b[i % period];
}

// ---- BiSeNet face parsing (models/CtrlHair/external_code/face_parsing/model.py, resnet.py) ----
// MaxPool2d(kernel 3, stride 2, padding 1) (resnet.py:62): padding behaves like -inf
__global__ __launch_bounds__(256) void maxpool3x3s2(float *__restrict__ out, const float *__restrict__ x, long long planes, int h,
                                                    int w, int oh, int ow) {
  const long long total = planes * oh * ow, stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int ox = (int)(i % ow), oy = (int)((i / ow) % oh);
    const float *src = x + (i / ((long long)oh * ow)) * h * w;
    float m = -3.402823466e38f;
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        const int iy = 2 * oy + dy, ix = 2 * ox + dx;
        if (iy >= 0 && iy < h && ix >= 0 && ix < w) m = fmaxf(m, src[(long long)iy * w + ix]);
      }
    out[i] = m;
  }
}

// out[p][i] = x[p][i] * (sigmoid(logit[p]) + plus_one) + add_plane[p][i] + add_bcast[p]
// (AttentionRefinementModule: feat * sigmoid(bn(conv(avgpool))) [+ the next stage's term], model.py:80-86, 111-122;
//  FeatureFusionModule: feat * atten + feat, model.py:200-207); add_* may be NULL
__global__ __launch_bounds__(256) void gate_kernel(float *__restrict__ out, const float *__restrict__ x,
                                                   const float *__restrict__ logit, const float *__restrict__ add_plane,
                                                   const float *__restrict__ add_bcast, float plus_one, long long planes, int hw) {
  const long long total = planes * hw, stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long long p = i / hw;
    const float g = 1.0f / (1.0f + expf(-logit[p])) + plus_one;
    float v = x[i] * g;
    if (add_plane) v += add_plane[i];
    if (add_bcast) v += add_bcast[p];
    out[i] = v;
  }
}

// F.interpolate(mode='nearest') to (oh, ow): src index = min(floor(dst * in / out), in - 1)
__global__ __launch_bounds__(256) void upsample_nearest(float *__restrict__ out, const float *__restrict__ x, long long planes,
                                                        int h, int w, int oh, int ow) {
  const long long total = planes * oh * ow, stride = (long long)gridDim.x * blockDim.x;
  const float sy = (float)h / (float)oh, sx = (float)w / (float)ow;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int ox = (int)(i % ow), oy = (int)((i / ow) % oh);
    const int iy = min((int)floorf(oy * sy), h - 1), ix = min((int)floorf(ox * sx), w - 1);
    out[i] = x[(i / ((long long)oh * ow)) * h * w + (long long)iy * w + ix];
  }
}

// get_segmentation's tail in one pass (model.py:249, my_parsing_util.py:86-95, models/Net.py:111-114): for every
// pixel of the (oh, ow) nearest-resized mask take its source pixel in the (H, W) full-resolution plane, evaluate the
// bilinear (align_corners=True) up-sampling of the `classes` logit planes [h, w] there - ATen's formula and operation
// order, no fused multiply-adds, so that equal inputs give equal values - take the FIRST maximum, apply the label
// permutation.  The [classes, H, W] logits and the [H, W] mask are never materialised.
__global__ __launch_bounds__(256) void parsing_mask(long long *__restrict__ out, const float *__restrict__ logits,
                                                    const int *__restrict__ remap, int classes, int h, int w, int H, int W,
                                                    int oh, int ow, long long total) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  const float ry = (H > 1) ? (float)(h - 1) / (float)(H - 1) : 0.0f, rx = (W > 1) ? (float)(w - 1) / (float)(W - 1) : 0.0f;
  const float ny = (float)H / (float)oh, nx = (float)W / (float)ow;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int ox = (int)(i % ow), oy = (int)((i / ow) % oh);
    const long long img = i / ((long long)oh * ow);
    const int Y = min((int)floorf(oy * ny), H - 1), X = min((int)floorf(ox * nx), W - 1);  // nearest resize
    const float fy = ry * Y, fx = rx * X;
    const int y0 = (int)fy, x0 = (int)fx;
    const int yp = (y0 < h - 1) ? 1 : 0, xp = (x0 < w - 1) ? 1 : 0;
    float ly1 = fy - (float)y0, lx1 = fx - (float)x0;
    float ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
    HF_OPAQUE_F32(ly0); HF_OPAQUE_F32(ly1); HF_OPAQUE_F32(lx0); HF_OPAQUE_F32(lx1);
    const float *base = logits + img * classes * h * w + (long long)y0 * w + x0;
    float best = 0.0f;
    int arg = 0;
    for (int c = 0; c < classes; ++c) {
      const float *pl = base + (long long)c * h * w;
      // h0lambda * (w0lambda * a + w1lambda * b) + h1lambda * (w0lambda * c + w1lambda * d), products and sums rounded separately
      float t0 = lx0 * pl[0], t1 = lx1 * pl[xp], t2 = lx0 * pl[yp * w], t3 = lx1 * pl[yp * w + xp];
      HF_OPAQUE_F32(t0); HF_OPAQUE_F32(t1); HF_OPAQUE_F32(t2); HF_OPAQUE_F32(t3);
      float top = t0 + t1, bot = t2 + t3;
      HF_OPAQUE_F32(top); HF_OPAQUE_F32(bot);
      float u0 = ly0 * top, u1 = ly1 * bot;
      HF_OPAQUE_F32(u0); HF_OPAQUE_F32(u1);
      const float v = u0 + u1;
      if (c == 0 || v > best) {
        best = v;
        arg = c;
      }
    }
    out[i] = remap ? remap[arg] : arg;
  }
}

// ---- the stencils on either side of the hot path ----
// BicubicDownSample.forward (utils/bicubic.py:38-75): reflect padding, then the separable 4*factor-tap filter k
// applied down the columns (stride factor) and along the rows (stride factor), fp32.  One thread per output pixel:
// the column sums of its 4*factor input columns first, then their weighted sum - the reference's order of the
// two 1-D passes.
constexpr int kBicMaxTaps = 32;
// One block per (plane, output row): the vertical pass of the row's w input columns goes to LDS (coalesced row
// loads, one column per thread and step), the horizontal pass reads it back - 4*factor loads per input column
// instead of (4*factor)^2 strided loads per output pixel (the one-thread-per-output form took 1.3 ms for the
// 24 images of a batched swap).  Same operation order as before: column sums first, then their weighted sum.
__global__ __launch_bounds__(256) void bicubic_down(float *__restrict__ out, const float *__restrict__ x,
                                                    const float *__restrict__ k1d, long long planes, int h, int w, int factor,
                                                    int oh, int ow) {
  HF_DYN_LDS;
  float *col = reinterpret_cast<float *>(hf_dyn_lds);  // [w]
  const int taps = 4 * factor, lo = (taps - factor) / 2;
  float k[kBicMaxTaps];
#pragma unroll
  for (int i = 0; i < kBicMaxTaps; ++i) k[i] = (i < taps) ? k1d[i] : 0.0f;
  for (long long row = blockIdx.x; row < planes * oh; row += gridDim.x) {
    const int oy = (int)(row % oh);
    const float *src = x + (row / oh) * h * w;
    for (int ix = threadIdx.x; ix < w; ix += blockDim.x) {
      float c = 0.0f;
      for (int i = 0; i < taps; ++i) {
        int iy = oy * factor + i - lo;
        iy = iy < 0 ? -iy : (iy >= h ? 2 * (h - 1) - iy : iy);  // F.pad(mode='reflect')
        c = fmaf(k[i], src[(long long)iy * w + ix], c);
      }
      col[ix] = c;
    }
    __syncthreads();
    for (int ox = threadIdx.x; ox < ow; ox += blockDim.x) {
      float acc = 0.0f;
      for (int j = 0; j < taps; ++j) {
        int ix = ox * factor + j - lo;
        ix = ix < 0 ? -ix : (ix >= w ? 2 * (w - 1) - ix : ix);
        acc = fmaf(k[j], col[ix], acc);
      }
      out[row * ow + ox] = acc;
    }
    __syncthreads();
  }
}

// DilateErosion.mask (utils/image_utils.py:42-55) on a BINARY mask: `radius` rounds of the 4-neighbourhood cross
// (dilation: any neighbour set; erosion: all five set, outside the image counts as unset) = one pass with the
// diamond |dy| + |dx| <= radius.
__global__ __launch_bounds__(256) void dilate_erode(float *__restrict__ dil, float *__restrict__ ero,
                                                    const float *__restrict__ mask, long long planes, int h, int w, int radius) {
  const long long total = planes * h * w, stride = (long long)gridDim.x * blockDim.x;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int xx = (int)(t % w), yy = (int)((t / w) % h);
    const float *src = mask + (t / ((long long)h * w)) * h * w;
    bool any = false, all = true;
    for (int dy = -radius; dy <= radius; ++dy) {
      const int r = radius - (dy < 0 ? -dy : dy);
      for (int dx = -r; dx <= r; ++dx) {
        const int y = yy + dy, x_ = xx + dx;
        const bool in = y >= 0 && y < h && x_ >= 0 && x_ < w;
        const bool set = in && src[(long long)y * w + x_] > 0.0f;
        any = any || set;
        all = all && set;
      }
    }
    dil[t] = any ? 1.0f : 0.0f;
    ero[t] = all ? 1.0f : 0.0f;
  }
}

__global__ __launch_bounds__(256) void axpby_bcast(float *__restrict__ out, const float *__restrict__ a, float alpha,
                                                   const float *__restrict__ b, float beta, long long n, long long period) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = fmaf(alpha, a[i], beta * b[i % period]);
}

inline int grid_for(long long n) {
  long long g = (n + 255) / 256;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" int hf_conv_prepare_f32(float *wt, const float *weight, int cout, int cin, int k, float scale,
                                   void *stream) {
  if (!wt || !weight || cout <= 0 || cin <= 0 || (k != 1 && k != 3 && k != 7)) return HF_E_INVALID;
  hipLaunchKernelGGL(conv_prepare, dim3(grid_for((long long)cout * cin)), dim3(256), 0, (hipStream_t)stream, wt,
                     weight, cout, cin, k * k, scale);
  return hf_launch_status();
}

extern "C" int hf_bn_fold_f32(float *scale, float *shift, const float *gamma, const float *beta, const float *mean,
                              const float *var, const float *conv_bias, float eps, int n, void *stream) {
  if (!scale || !shift || !gamma || !beta || !mean || !var || n <= 0) return HF_E_INVALID;
  hipLaunchKernelGGL(bn_fold, dim3(hf_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, scale, shift, gamma, beta,
                     mean, var, conv_bias, eps, n);
  return hf_launch_status();
}

extern "C" int hf_plane_mean_f32(float *out, const float *x, int planes, int hw, void *stream) {
  if (!out || !x || planes <= 0 || hw <= 0) return HF_E_INVALID;
  if (hw >= 4096 && (hw & 3) == 0 && planes <= 65535 * 32)
    hipLaunchKernelGGL(plane_mean_block, dim3(planes), dim3(256), 4 * sizeof(float), (hipStream_t)stream, out, x, hw);
  else
    hipLaunchKernelGGL(plane_mean, dim3(hf_cdiv(planes, 4)), dim3(256), 0, (hipStream_t)stream, out, x, planes, hw);
  return hf_launch_status();
}

extern "C" int hf_se_gate_f32(float *gate, const float *pooled, const float *fc1, const float *fc2, int batch,
                              int channels, int reduced, void *stream) {
  if (!gate || !pooled || !fc1 || !fc2 || batch <= 0 || channels <= 0 || reduced <= 0) return HF_E_INVALID;
  const size_t lds = (size_t)(channels + reduced) * sizeof(float);
  if (lds > 64 * 1024) return HF_E_INVALID;
  hipLaunchKernelGGL(se_gate, dim3(batch), dim3(256), lds, (hipStream_t)stream, gate, pooled, fc1, fc2, channels,
                     reduced);
  return hf_launch_status();
}

extern "C" int hf_scale_shortcut_add_f32(float *out, const float *r, const float *gate, const float *shortcut,
                                         int sc_stride, int batch, int channels, int oh, int ow, int sh, int sw,
                                         void *stream) {
  if (!out || !r || !shortcut || sc_stride < 1 || batch <= 0 || channels <= 0 || oh <= 0 || ow <= 0) return HF_E_INVALID;
  if ((oh - 1) * sc_stride >= sh || (ow - 1) * sc_stride >= sw) return HF_E_INVALID;
  const long long total = (long long)batch * channels * oh * ow;
  hipLaunchKernelGGL(scale_shortcut_add, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, out, r, gate,
                     shortcut, sc_stride, oh, ow, sh, sw, total);
  return hf_launch_status();
}

extern "C" int hf_upsample_bilinear_add_f32(float *out, const float *x, const float *y, int planes, int h, int w,
                                            int oh, int ow, void *stream) {
  if (!out || !x || !y || planes <= 0 || h <= 0 || w <= 0 || oh <= 0 || ow <= 0) return HF_E_INVALID;
  const long long total = (long long)planes * oh * ow;
  hipLaunchKernelGGL(upsample_bilinear_add, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, out, x, y, h,
                     w, oh, ow, total);
  return hf_launch_status();
}

extern "C" int hf_adaptive_avgpool_f32(float *out, const float *x, int batch, int channels, int h, int w, int oh,
                                       int ow, int out_channels_total, int out_channel_offset, void *stream) {
  if (!out || !x || batch <= 0 || channels <= 0 || h <= 0 || w <= 0 || oh <= 0 || ow <= 0 ||
      out_channel_offset < 0 || out_channel_offset + channels > out_channels_total)
    return HF_E_INVALID;
  const long long nitems = (long long)batch * channels * oh * ow;
  hipLaunchKernelGGL(adaptive_avgpool, dim3(hf_cdiv(nitems, 4)), dim3(256), 0, (hipStream_t)stream, out, x, batch,
                     channels, h, w, oh, ow, out_channels_total, out_channel_offset);
  return hf_launch_status();
}

extern "C" int hf_downscale2x_f32(float *out, const float *x, int planes, int h, int w, void *stream) {
  if (!out || !x || planes <= 0 || h < 2 || w < 2 || (h & 1) || (w & 1)) return HF_E_INVALID;
  const long long total = (long long)planes * (h / 2) * (w / 2);
  hipLaunchKernelGGL(downscale2x, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, out, x, h, w, total);
  return hf_launch_status();
}

static void launch_linear(float *out, const float *x, long long x_stride, const float *w, const float *bias, int batch, int in_f,
                          int out_f, float scale, float bias_scale, int act, float alpha, float act_scale, hipStream_t st) {
  // the K-split form issues 16-byte loads of x and w: base pointers must be 16-byte aligned too (a view with a storage
  // offset - x[:, 1:] re-strided, a weight slice - takes the scalar-tolerant kernel below)
  const bool aligned16 = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 15) == 0;
  if (in_f >= 4096 && (in_f & 15) == 0 && (x_stride & 3) == 0 && aligned16)
    hipLaunchKernelGGL(linear_kernel_ksplit, dim3(hf_cdiv(out_f, kLinRows), hf_cdiv(batch, kLinMaxBatch)), dim3(256),
                       4 * kLinRows * kLinMaxBatch * sizeof(float), st, out, x, x_stride, w, bias, batch, in_f, out_f, scale,
                       bias_scale, act, alpha, act_scale);
  else
    hipLaunchKernelGGL(linear_kernel, dim3(hf_cdiv(out_f, 4 * kLinRows), hf_cdiv(batch, kLinMaxBatch)), dim3(256), 0, st, out, x,
                       x_stride, w, bias, batch, in_f, out_f, scale, bias_scale, act, alpha, act_scale);
}

extern "C" int hf_linear_f32(float *out, const float *x, long long x_stride, const float *w, const float *bias,
                             int batch, int in_features, int out_features, float scale, void *stream) {
  if (!out || !x || !w || batch <= 0 || batch > 65535 * kLinMaxBatch || in_features <= 0 || out_features <= 0)
    return HF_E_INVALID;
  launch_linear(out, x, x_stride, w, bias, batch, in_features, out_features, scale, 1.0f, 0, 0.0f, 1.0f, (hipStream_t)stream);
  return hf_launch_status();
}

extern "C" int hf_equal_linear_f32(float *out, const float *x, long long x_stride, const float *w, const float *bias,
                                   int batch, int in_features, int out_features, float lr_mul, int fused_lrelu,
                                   float alpha, float act_scale, void *stream) {
  if (!out || !x || !w || batch <= 0 || batch > 65535 * kLinMaxBatch || in_features <= 0 || out_features <= 0)
    return HF_E_INVALID;
  const float scale = (1.0f / sqrtf((float)in_features)) * lr_mul;
  hipLaunchKernelGGL(linear_kernel, dim3(hf_cdiv(out_features, 4 * kLinRows), hf_cdiv(batch, kLinMaxBatch)), dim3(256), 0,
                     (hipStream_t)stream, out, x, x_stride, w, bias, batch, in_features, out_features, scale, lr_mul,
                     fused_lrelu ? 1 : 0, alpha, act_scale);
  return hf_launch_status();
}

extern "C" int hf_pixel_norm_f32(float *out, const float *x, int rows, int dim, void *stream) {
  if (!out || !x || rows <= 0 || dim <= 0) return HF_E_INVALID;
  hipLaunchKernelGGL(pixel_norm_rows, dim3(hf_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, out, x, rows, dim);
  return hf_launch_status();
}

extern "C" int hf_layernorm_f32(float *out, const float *x, const float *gamma, const float *beta, int rows, int dim, float eps,
                                int lrelu, float alpha, void *stream) {
  if (!out || !x || rows <= 0 || dim <= 0) return HF_E_INVALID;
  hipLaunchKernelGGL(layernorm_rows, dim3(rows), dim3(256), 32, (hipStream_t)stream, out, x, gamma, beta, dim, eps, lrelu ? 1 : 0,
                     alpha, 1);
  return hf_launch_status();
}

extern "C" int hf_layernorm_grouped_f32(float *out, const float *x, const float *gamma, const float *beta, int rows, int dim,
                                        int groups, float eps, int lrelu, float alpha, void *stream) {
  if (!out || !x || rows <= 0 || dim <= 0 || groups <= 0 || (rows % groups)) return HF_E_INVALID;
  hipLaunchKernelGGL(layernorm_rows, dim3(rows), dim3(256), 32, (hipStream_t)stream, out, x, gamma, beta, dim, eps, lrelu ? 1 : 0,
                     alpha, groups);
  return hf_launch_status();
}

// MUNIT-style LayerNorm of the CtrlHair shape adaptor (models/CtrlHair/my_torchlib/module.py:181-206): per SAMPLE
// mean and UNBIASED standard deviation over all C*H*W values, y = (x - mean) / (std + eps) * gamma[c] + beta[c], then
// LeakyReLU(slope) (slope 1 = none).  Three launches: (1) one block per (sample, chunk of 16384 values) computes the
// chunk's (mean, M2 = sum of squared deviations from it), its values read once and held in registers; (2) one thread per
// sample merges the chunk statistics in chunk order (Chan's pairwise update: exact counts, fixed order - deterministic);
// (3) the apply pass normalises.  (A single block per sample took 370 us on a 32 x 128^2 activation.)
constexpr int kLnChunk = 16384;
// VEC = 4: 16-byte loads (the launcher checks n % 4 == 0 and the alignment of x and of the sample stride)
template <int VEC>
__global__ __launch_bounds__(256) void sample_ln_partial(float *__restrict__ stats, const float *__restrict__ x, long long n,
                                                         long long x_bstride, int nchunks) {
  HF_DYN_LDS;
  float *red = reinterpret_cast<float *>(hf_dyn_lds);  // [8]
  const int b = blockIdx.y, k = blockIdx.x;
  const long long lo = (long long)k * kLnChunk, hi = lo + kLnChunk < n ? lo + kLnChunk : n;
  const float *xb = x + (long long)b * x_bstride;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // the thread's values stay in registers between the two passes (a chunk is 64 values per thread)
  constexpr int PER = kLnChunk / 256;
  float v[PER];
  float s = 0.0f;
  if (VEC == 4) {
#pragma unroll
    for (int j = 0; j < PER / 4; ++j) {
      const long long i = lo + ((long long)j * 256 + threadIdx.x) * 4;
      float4 q = make_float4(0, 0, 0, 0);
      if (i < hi) q = *reinterpret_cast<const float4 *>(xb + i);
      v[4 * j] = q.x, v[4 * j + 1] = q.y, v[4 * j + 2] = q.z, v[4 * j + 3] = q.w;
    }
  } else {
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const long long i = lo + (long long)j * 256 + threadIdx.x;
      v[j] = i < hi ? xb[i] : 0.0f;
    }
  }
#pragma unroll
  for (int j = 0; j < PER; ++j) s += v[j];
  s = hf_wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / (float)(hi - lo);
  float q = 0.0f;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const long long i = VEC == 4 ? lo + ((long long)(j / 4) * 256 + threadIdx.x) * 4 + (j & 3) : lo + (long long)j * 256 + threadIdx.x;
    const float d = v[j] - mean;
    if (i < hi) q = fmaf(d, d, q);
  }
  q = hf_wave_sum(q);
  if (lane == 0) red[4 + wave] = q;
  __syncthreads();
  if (threadIdx.x == 0) {
    stats[((long long)b * nchunks + k) * 2] = mean;
    stats[((long long)b * nchunks + k) * 2 + 1] = (red[4] + red[5]) + (red[6] + red[7]);
  }
}
// the chunk statistics of a sample merged in chunk order -> (mean, 1 / (std + eps)) in final[b]; one thread per sample
// (every block of the apply pass used to repeat this merge: 256 dependent updates, each with a division, in front of
// four values per thread - three quarters of the pass at 64 x 256^2)
__global__ void sample_ln_merge(float *__restrict__ final, const float *__restrict__ stats, long long n, int nchunks, int batch,
                                float eps) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  float mean = 0.0f, m2 = 0.0f, cnt = 0.0f;
  for (int k = 0; k < nchunks; ++k) {
    const long long lo = (long long)k * kLnChunk;
    const float nk = (float)((lo + kLnChunk < n ? lo + kLnChunk : n) - lo);
    const float mk = stats[((long long)b * nchunks + k) * 2], qk = stats[((long long)b * nchunks + k) * 2 + 1];
    const float tot = cnt + nk, delta = mk - mean;
    mean += delta * (nk / tot);
    m2 += qk + delta * delta * (cnt * nk / tot);
    cnt = tot;
  }
  final[2 * b] = mean;
  final[2 * b + 1] = 1.0f / (sqrtf(m2 / (float)(n - 1)) + eps);
}
template <int VEC>
__global__ __launch_bounds__(256) void sample_ln_apply(float *__restrict__ out, const float *__restrict__ x,
                                                       const float *__restrict__ final, const float *__restrict__ gamma,
                                                       const float *__restrict__ beta, long long n, long long x_bstride, int hw,
                                                       float slope) {
  const int b = blockIdx.y;
  const float mean = final[2 * b], inv = final[2 * b + 1];
  const float *xb = x + (long long)b * x_bstride;
  float *ob = out + (long long)b * n;
  const long long stride = (long long)gridDim.x * 256 * VEC;
  for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * VEC; i < n; i += stride) {
    const int c = (int)(i / hw);  // VEC == 4: hw % 4 == 0, the four values share the channel
    const float g = gamma ? gamma[c] : 1.0f, bt = (gamma && beta) ? beta[c] : 0.0f;
    if (VEC == 4) {
      const float4 q = *reinterpret_cast<const float4 *>(xb + i);
      float y[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        y[j] = (y[j] - mean) * inv;
        if (gamma) y[j] = fmaf(y[j], g, bt);
        y[j] = y[j] > 0.0f ? y[j] : y[j] * slope;
      }
      *reinterpret_cast<float4 *>(ob + i) = make_float4(y[0], y[1], y[2], y[3]);
    } else {
      float y = (xb[i] - mean) * inv;
      if (gamma) y = fmaf(y, g, bt);
      ob[i] = y > 0.0f ? y : y * slope;
    }
  }
}

extern "C" long long hf_sample_layernorm_workspace_floats(int batch, int channels, int hw) {
  const long long n = (long long)channels * hw;
  return batch <= 0 || n <= 0 ? 0 : 2LL * batch * ((n + kLnChunk - 1) / kLnChunk + 1);
}

extern "C" int hf_sample_layernorm_f32(float *out, const float *x, const float *gamma, const float *beta, int batch, int channels,
                                       int hw, long long x_batch_stride, float eps, float slope, float *workspace,
                                       long long workspace_floats, void *stream) {
  const long long n = (long long)channels * hw;
  if (x_batch_stride == 0) x_batch_stride = n;
  if (!out || !x || batch <= 0 || channels <= 0 || hw <= 0 || n < 2 || batch > 65535 || x_batch_stride < n) return HF_E_INVALID;
  const int nchunks = (int)((n + kLnChunk - 1) / kLnChunk);
  if (!workspace || workspace_floats < 2LL * batch * (nchunks + 1)) return HF_E_WORKSPACE;
  float *final = workspace + 2LL * batch * nchunks;
  const bool vec = (hw & 3) == 0 && (x_batch_stride & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  hipStream_t st = (hipStream_t)stream;
  long long blocks = (n + 4095) / 4096;  // 16 values per thread
  if (blocks > 4096) blocks = 4096;
  if (vec) hipLaunchKernelGGL(sample_ln_partial<4>, dim3(nchunks, batch), dim3(256), 8 * sizeof(float), st, workspace, x, n, x_batch_stride, nchunks);
  else hipLaunchKernelGGL(sample_ln_partial<1>, dim3(nchunks, batch), dim3(256), 8 * sizeof(float), st, workspace, x, n, x_batch_stride, nchunks);
  hipLaunchKernelGGL(sample_ln_merge, dim3(hf_cdiv(batch, 64)), dim3(64), 0, st, final, workspace, n, nchunks, batch, eps);
  if (vec) hipLaunchKernelGGL(sample_ln_apply<4>, dim3((int)blocks, batch), dim3(256), 0, st, out, x, final, gamma, beta, n, x_batch_stride, hw, slope);
  else hipLaunchKernelGGL(sample_ln_apply<1>, dim3((int)blocks, batch), dim3(256), 0, st, out, x, final, gamma, beta, n, x_batch_stride, hw, slope);
  return hf_launch_status();
}

extern "C" int hf_modulate_f32(float *out, const float *x, const float *gamma, const float *beta, long long n, int lrelu,
                               float alpha, void *stream) {
  if (!out || !x || !gamma || !beta || n <= 0) return HF_E_INVALID;
  hipLaunchKernelGGL(modulate_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, out, x, gamma, beta, n, lrelu ? 1 : 0,
                     alpha);
  return hf_launch_status();
}

extern "C" int hf_pixel_norm_dim1_f32(float *out, const float *x, int batch, int layers, int dim, void *stream) {
  if (!out || !x || batch <= 0 || layers <= 0 || dim <= 0) return HF_E_INVALID;
  hipLaunchKernelGGL(pixel_norm_dim1, dim3(hf_cdiv((long long)batch * dim, 256)), dim3(256), 0, (hipStream_t)stream, out, x, batch,
                     layers, dim);
  return hf_launch_status();
}

extern "C" int hf_maxpool3x3s2_f32(float *out, const float *x, long long planes, int h, int w, void *stream) {
  if (!out || !x || planes <= 0 || h <= 0 || w <= 0) return HF_E_INVALID;
  const int oh = (h - 1) / 2 + 1, ow = (w - 1) / 2 + 1;
  hipLaunchKernelGGL(maxpool3x3s2, dim3(grid_for(planes * oh * ow)), dim3(256), 0, (hipStream_t)stream, out, x, planes, h, w, oh, ow);
  return hf_launch_status();
}

extern "C" int hf_gate_f32(float *out, const float *x, const float *logit, const float *add_plane, const float *add_bcast,
                           float plus_one, long long planes, int hw, void *stream) {
  if (!out || !x || !logit || planes <= 0 || hw <= 0) return HF_E_INVALID;
  hipLaunchKernelGGL(gate_kernel, dim3(grid_for(planes * hw)), dim3(256), 0, (hipStream_t)stream, out, x, logit, add_plane,
                     add_bcast, plus_one, planes, hw);
  return hf_launch_status();
}

extern "C" int hf_upsample_nearest_f32(float *out, const float *x, long long planes, int h, int w, int oh, int ow, void *stream) {
  if (!out || !x || planes <= 0 || h <= 0 || w <= 0 || oh <= 0 || ow <= 0) return HF_E_INVALID;
  hipLaunchKernelGGL(upsample_nearest, dim3(grid_for(planes * oh * ow)), dim3(256), 0, (hipStream_t)stream, out, x, planes, h, w,
                     oh, ow);
  return hf_launch_status();
}

extern "C" int hf_parsing_mask_i64(long long *out, const float *logits, const int *remap, int images, int classes, int h, int w,
                                   int full_h, int full_w, int out_h, int out_w, void *stream) {
  if (!out || !logits || images <= 0 || classes <= 0 || h <= 0 || w <= 0 || full_h <= 0 || full_w <= 0 || out_h <= 0 || out_w <= 0)
    return HF_E_INVALID;
  const long long total = (long long)images * out_h * out_w;
  hipLaunchKernelGGL(parsing_mask, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, out, logits, remap, classes, h, w,
                     full_h, full_w, out_h, out_w, total);
  return hf_launch_status();
}

extern "C" int hf_bicubic_down_f32(float *out, const float *x, const float *k1d, long long planes, int h, int w, int factor,
                                   void *stream) {
  if (!out || !x || !k1d || planes <= 0 || factor < 1 || 4 * factor > kBicMaxTaps || h % factor || w % factor ||
      h < 4 * factor || w < 4 * factor)
    return HF_E_INVALID;
  const int oh = h / factor, ow = w / factor;
  if (w > 16384) return HF_E_INVALID;  // one input row of column sums in LDS
  const long long rows = planes * oh;
  hipLaunchKernelGGL(bicubic_down, dim3((int)(rows > 65536 ? 65536 : rows)), dim3(256), (size_t)w * sizeof(float), (hipStream_t)stream,
                     out, x, k1d, planes, h, w, factor, oh, ow);
  return hf_launch_status();
}

extern "C" int hf_dilate_erode_f32(float *dilated, float *eroded, const float *mask, long long planes, int h, int w, int radius,
                                   void *stream) {
  if (!dilated || !eroded || !mask || planes <= 0 || h <= 0 || w <= 0 || radius < 0 || radius > 64) return HF_E_INVALID;
  hipLaunchKernelGGL(dilate_erode, dim3(grid_for(planes * h * w)), dim3(256), 0, (hipStream_t)stream, dilated, eroded, mask, planes,
                     h, w, radius);
  return hf_launch_status();
}

extern "C" int hf_axpby_bcast_f32(float *out, const float *a, float alpha, const float *b, float beta, long long n,
                                  long long b_period, void *stream) {
  if (!out || !a || !b || n <= 0 || b_period <= 0) return HF_E_INVALID;
  hipLaunchKernelGGL(axpby_bcast, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, out, a, alpha, b, beta, n, b_period);
  return hf_launch_status();
}

extern "C" int hf_add_bcast_f32(float *out, const float *a, const float *b, long long n, long long b_period,
                                void *stream) {
  if (!out || !a || !b || n <= 0 || b_period <= 0) return HF_E_INVALID;
  hipLaunchKernelGGL(add_bcast, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, out, a, b, n, b_period);
  return hf_launch_status();
}
