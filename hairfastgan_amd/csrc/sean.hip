// sean.hip - kernels of the SEAN inpainting stage (models/sean_codes/models/networks/{generator,architecture,
// normalization}.py).  The dense 3x3 convolutions run on the library's conv kernels
// (csrc/convh_enc.hip / modconv.hip); what is here is what makes SEAN's normalisation cheap on this machine:
//
//  * A 3x3 convolution whose input is PIECEWISE CONSTANT per segmentation label - SPADE's mlp_shared on the one-hot
//    map (normalization.py:246-252), ACE's conv_gamma / conv_beta on `middle_avg` (the per-region style vector
//    broadcast over the region, :117-160), the generator's `fc` on the down-sampled one-hot map (generator.py:75-76) -
//    is a sum of nine table entries: out[c, p] = bias[c] + sum_tap T[tap, c, label(p + tap)], T = W_tap . v_label.
//    ACE's two 512 -> C convolutions (2 x 77 GFLOP at 128^2 for one image) become a 19-row GEMM (the table) plus nine
//    gathers per output element.
//  * ACE's tail in one pass: noise, inference BatchNorm, the gamma / beta blend, (1 + gamma) * x + beta, LeakyReLU.
//  * The style encoder's per-region average pooling (architecture.py:187-205) with the final tanh folded in.
#include "hf_common.h"

// Floating-point contraction: "on" = a multiply and an add are fused only where they are written in ONE expression (or as
// fmaf), never across statements.  hipcc's default ("fast") lets the backend fuse opportunistically per basic block: the
// unrolled body of a grid-stride loop and its remainder iterations then round differently, i.e. a sample's bits depend on
// how many elements the launch has - on what it is batched with (found by tools/probes/batch_variance.py in
// upsample_bilinear_add: the third of three images differed from the third of six by one ulp).
#pragma clang fp contract(on)

constexpr int kSeanLabels = 19;

#ifndef HF_WAVE_ANY_DEFINED
#define HF_WAVE_ANY_DEFINED
__device__ __forceinline__ bool hf_wave_any(bool p) { return __any((int)p) != 0; }
#endif

// out[b, c, y, x] = act( bias[c] + sum_{tap=(ky,kx)} table[(tap*C + c) * tcols + col0(b) + label[b / group, y+ky-1, x+kx-1]] )
// taps outside the image contribute nothing (zero padding).  table: [9*C][tcols]; col0(b) = b * cols_per_sample.
// tsum [C][tcols] = sum over the nine taps of table (NULL = off): a pixel whose 3x3 neighbourhood lies inside the image
// and carries ONE label - the interior of a region, most pixels at the high resolutions - needs one lookup instead of nine;
// a wave takes that path when all of its 64 pixels qualify.  (The interior sum is the table's taps added in a fixed order
// by the caller, the general path adds them in tap order: the two paths differ by an fp32 reassociation.)
__global__ __launch_bounds__(256) void label_conv3x3(float *__restrict__ out, const int *__restrict__ labels,
                                                     const float *__restrict__ table, const float *__restrict__ tsum,
                                                     const float *__restrict__ bias, int C, int H, int W, int tcols,
                                                     int cols_per_sample, int group, int act, int cchunk) {
  const int hw = H * W;
  const int p = blockIdx.x * 256 + threadIdx.x;
  const bool live = p < hw;
  const int b = blockIdx.z;
  const int pc = live ? p : hw - 1;
  const int y = pc / W, x = pc - y * W;
  const int *lb = labels + (long long)(b / group) * hw;
  const int nl = cols_per_sample > 0 ? cols_per_sample : tcols;  // labels per sample
  int col[9];
  bool uniform = tsum != nullptr;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
    const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
    // labels outside [0, nl) (an 'ignore' value) are clamped as in label_window: both kernels agree, no read outside the sample's columns
    col[t] = in ? b * cols_per_sample + min(max(lb[yy * W + xx], 0), nl - 1) : -1;
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) uniform = uniform && col[t] == col[4];
  const int c0 = blockIdx.y * cchunk, c1 = min(C, c0 + cchunk);
  float *o = out + ((long long)b * C + c0) * hw + pc;
  if (!hf_wave_any(!uniform)) {
    for (int c = c0; c < c1; ++c, o += hw) {
      const float acc = (bias ? bias[c] : 0.0f) + tsum[(long long)c * tcols + col[4]];
      if (live) *o = act ? fmaxf(acc, 0.0f) : acc;
    }
    return;
  }
  for (int c = c0; c < c1; ++c, o += hw) {
    float acc = bias ? bias[c] : 0.0f;
#pragma unroll
    for (int t = 0; t < 9; ++t)
      if (col[t] >= 0) acc += table[((long long)t * C + c) * tcols + col[t]];
    if (live) *o = act ? fmaxf(acc, 0.0f) : acc;
  }
}

// tsum[c][j] = sum_tap table[tap*C + c][j]  (taps added in order 0..8)
__global__ __launch_bounds__(256) void label_table_sum(float *__restrict__ tsum, const float *__restrict__ table, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float acc = 0.0f;
#pragma unroll
  for (int t = 0; t < 9; ++t) acc += table[(long long)t * n + i];
  tsum[i] = acc;
}

// ---- four pixels per thread, tables in LDS (planes of >= 1024 pixels, W % 4 == 0, <= 32 labels) ----------------------
// The per-pixel form above is bound by its gathers (one L2 lookup and one 256-byte store per wave and channel: 0.2 TB/s
// of output at 128^2 x 512 channels); here a block first copies the nine tap tables of its channel chunk and of ITS
// sample's columns into LDS (and adds them up, taps 0..8 in order, for the interior path), a thread then owns four
// consecutive pixels of a row and writes 16 bytes per channel.  The interior / general choice is made PER PIXEL (a
// pixel's value does not depend on which pixels share its wave): interior = bias + tap sum, general = bias + taps in
// order; waves without a general pixel skip that arithmetic.
constexpr int kLabelMax = 32;

// labels of rows y-1..y+1, columns x-1..x+4 around the thread's four pixels (x % 4 == 0): -1 outside the image,
// clamped to [0, nl) inside
__device__ __forceinline__ void label_window(int (&lab)[3][6], const int *__restrict__ lb, int y, int x, int H, int W, int nl) {
  auto in_range = [nl](int v) { return min(max(v, 0), nl - 1); };
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int yy = y + r - 1;
    if (yy >= 0 && yy < H) {
      const int *row = lb + (long long)yy * W + x;
      const int4 c = *reinterpret_cast<const int4 *>(row);
      lab[r][0] = x > 0 ? in_range(row[-1]) : -1;
      lab[r][1] = in_range(c.x);
      lab[r][2] = in_range(c.y);
      lab[r][3] = in_range(c.z);
      lab[r][4] = in_range(c.w);
      lab[r][5] = x + 4 < W ? in_range(row[4]) : -1;
    } else {
#pragma unroll
      for (int j = 0; j < 6; ++j) lab[r][j] = -1;
    }
  }
}

// does pixel j's 3x3 neighbourhood lie inside the image and carry one label
__device__ __forceinline__ bool label_interior(const int (&lab)[3][6], int j) {
  bool u = true;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int k = 0; k < 3; ++k) u = u && lab[r][j + k] == lab[1][j + 1];
  return u;
}

// LDS tables of `rows` table rows per tap: tl[(tap*rows + i)*nl + label] and ts[i*nl + label] = their sum over the taps
// (0 + tap 0 + ... + tap 8, the order of label_table_sum).  row_of(i) = the table row (channel) of local row i.
template <class RowOf>
__device__ __forceinline__ void stage_label_tables(float *tl, float *ts, const float *__restrict__ table, int C, int tcols, int col0,
                                                   int nl, int rows, RowOf row_of) {
  for (int i = threadIdx.x; i < 9 * rows * nl; i += 256) {
    const int t = i / (rows * nl), rem = i - t * rows * nl;
    const int r = rem / nl, l = rem - r * nl;
    tl[i] = table[((long long)t * C + row_of(r)) * tcols + col0 + l];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < rows * nl; i += 256) {
    float acc = 0.0f;
#pragma unroll
    for (int t = 0; t < 9; ++t) acc += tl[t * rows * nl + i];
    ts[i] = acc;
  }
  __syncthreads();
}

// value of local table row i at the thread's four pixels: bias + (interior ? tap sum : taps in order)
__device__ __forceinline__ void label_lookup4(float (&v)[4], const float *tl, const float *ts, int rows, int nl, int i, float bias,
                                              const int (&lab)[3][6], const bool (&inner)[4], bool any_general) {
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = bias + ts[i * nl + lab[1][j + 1]];
  if (any_general) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float acc = bias;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int l = lab[t / 3][j + t % 3];
        if (l >= 0) acc += tl[(t * rows + i) * nl + l];
      }
      if (!inner[j]) v[j] = acc;
    }
  }
}

__global__ __launch_bounds__(256) void label_conv3x3_v4(float *__restrict__ out, const int *__restrict__ labels,
                                                        const float *__restrict__ table, const float *__restrict__ bias, int C,
                                                        int H, int W, int tcols, int cols_per_sample, int group, int act,
                                                        int cchunk, int nl, int interior) {
  HF_DYN_LDS;
  float *tl = reinterpret_cast<float *>(hf_dyn_lds);  // [9][nc][nl]
  const int b = blockIdx.z, c0 = blockIdx.y * cchunk, nc = min(cchunk, C - c0);
  float *ts = tl + 9 * cchunk * nl;                   // [nc][nl]
  stage_label_tables(tl, ts, table, C, tcols, b * cols_per_sample, nl, nc, [&](int r) { return c0 + r; });
  const int hw = H * W;
  const int p = (blockIdx.x * 256 + threadIdx.x) * 4;
  const bool live = p < hw;
  const int pc = live ? p : 0;
  const int y = pc / W, x = pc - y * W;
  int lab[3][6];
  label_window(lab, labels + (long long)(b / group) * hw, y, x, H, W, nl);
  bool inner[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) inner[j] = interior && label_interior(lab, j);
  const bool any_general = hf_wave_any(live && !(inner[0] && inner[1] && inner[2] && inner[3]));
  float *o = out + ((long long)b * C + c0) * hw + pc;
  for (int c = 0; c < nc; ++c, o += hw) {
    float v[4];
    label_lookup4(v, tl, ts, nc, nl, c, bias ? bias[c0 + c] : 0.0f, lab, inner, any_general);
    if (act) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.0f);
    }
    if (live) *reinterpret_cast<float4 *>(o) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

extern "C" int hf_label_conv3x3_f32(float *out, const int *labels, const float *table, const float *bias, int batch,
                                    int channels, int h, int w, int table_cols, int cols_per_sample, int group, int relu,
                                    float *tap_sum_scratch, void *stream) {
  if (!out || !labels || !table || batch <= 0 || channels <= 0 || h <= 0 || w <= 0 || table_cols <= 0 || group <= 0 ||
      cols_per_sample < 0 || batch > 65535)
    return HF_E_INVALID;
  const int hw = h * w;
  const int nl = cols_per_sample > 0 ? cols_per_sample : table_cols;  // labels per sample
  if (hw >= 1024 && (w & 3) == 0 && nl <= kLabelMax) {
    // channel chunk: enough blocks to fill the chip, tables of a block <= 10 * 64 * 32 floats (80 KiB)
    int cchunk = min(channels, 64);
    while (cchunk > 8 && (long long)hf_cdiv(hw, 1024) * hf_cdiv(channels, cchunk) * batch < 1024) cchunk = (cchunk + 1) / 2;
    const size_t lds = (size_t)10 * cchunk * nl * sizeof(float);
    hipLaunchKernelGGL(label_conv3x3_v4, dim3(hf_cdiv(hw, 1024), hf_cdiv(channels, cchunk), batch), dim3(256), lds,
                       (hipStream_t)stream, out, labels, table, bias, channels, h, w, table_cols, cols_per_sample, group,
                       relu ? 1 : 0, cchunk, nl, tap_sum_scratch ? 1 : 0);
    return hf_launch_status();
  }
  // enough blocks to fill the chip, at least 8 channels per thread to amortise the nine label loads
  int cchunk = channels;
  while (cchunk > 8 && (long long)hf_cdiv(hw, 256) * hf_cdiv(channels, cchunk) * batch < 1024) cchunk = (cchunk + 1) / 2;
  if (tap_sum_scratch) {
    const long long n = (long long)channels * table_cols;
    hipLaunchKernelGGL(label_table_sum, dim3(hf_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, tap_sum_scratch, table, n);
  }
  hipLaunchKernelGGL(label_conv3x3, dim3(hf_cdiv(hw, 256), hf_cdiv(channels, cchunk), batch), dim3(256), 0,
                     (hipStream_t)stream, out, labels, table, tap_sum_scratch, bias, channels, h, w, table_cols,
                     cols_per_sample, group, relu ? 1 : 0, cchunk);
  return hf_launch_status();
}

// ACE.forward's tail (normalization.py:106-107, 164-178):
//   n   = ((x + r[b,p] * noise_var[c]) - mean[c]) * rsqrt(var[c] + eps)            as n = (x + r*nv) * bn_scale + bn_shift
//   g   = a_g * avg[b, c] + (1 - a_g) * sp[b / group, c],     a_g = sigmoid(blend[0])    (avg NULL: g = sp)
//   bt  = a_b * avg[b, C + c] + (1 - a_b) * sp[b / group, C + c], a_b = sigmoid(blend[1])
//   out = lrelu_slope( n * (1 + g) + bt )                                           (slope 1: no activation)
// avg / sp: [*, 2C, H, W] (gamma planes first, then beta planes).
// one element of the tail, every product-sum an explicit fma: ace_modulate and ace_modulate_table must agree bit for bit
// (left to the compiler, the two kernels contracted a * b + c * d differently)
__device__ __forceinline__ float ace_tail(float x, float r, float nv, float sc, float sh, float g, float bt, float slope) {
  const float n = fmaf(fmaf(r, nv, x), sc, sh);
  const float v = fmaf(n, 1.0f + g, bt);
  return v > 0.0f ? v : v * slope;
}
__device__ __forceinline__ float ace_blend(float a, float avg, float spv) { return fmaf(a, avg, (1.0f - a) * spv); }

__global__ __launch_bounds__(256) void ace_modulate(float *__restrict__ out, const float *__restrict__ x,
                                                    const float *__restrict__ r, const float *__restrict__ noise_var,
                                                    const float *__restrict__ bn_scale, const float *__restrict__ bn_shift,
                                                    const float *__restrict__ avg, const float *__restrict__ sp,
                                                    const float *__restrict__ blend, int C, int hw4, int group, float slope,
                                                    int w, int x_up) {
  const int c = blockIdx.y, b = blockIdx.z;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= hw4) return;
  const long long plane = (long long)hw4;  // in float4 units
  float4 xv;
  if (x_up) {  // x is the HALF-resolution plane [h/2, w/2]: nearest-neighbour x2 (generator.py:80-103 `self.up`) read in place
    const int p = i * 4, y = p / w, xx = p - y * w;
    const float2 lo = *reinterpret_cast<const float2 *>(x + ((long long)b * C + c) * plane + (long long)(y >> 1) * (w >> 1) + (xx >> 1));  // low-res plane = hw/4 = `plane` floats
    xv = make_float4(lo.x, lo.x, lo.y, lo.y);
  } else {
    xv = reinterpret_cast<const float4 *>(x)[((long long)b * C + c) * plane + i];
  }
  float4 rv = make_float4(0, 0, 0, 0);
  const float nv = noise_var ? noise_var[c] : 0.0f;
  if (r) rv = reinterpret_cast<const float4 *>(r)[(long long)b * plane + i];
  const float sc = bn_scale[c], sh = bn_shift[c];
  const long long sb = (long long)(b / group) * 2 * C;
  const float4 gs = reinterpret_cast<const float4 *>(sp)[(sb + c) * plane + i];
  const float4 bs = reinterpret_cast<const float4 *>(sp)[(sb + C + c) * plane + i];
  float4 g = gs, bt = bs;
  if (avg) {
    const float ag = 1.0f / (1.0f + expf(-blend[0])), ab = 1.0f / (1.0f + expf(-blend[1]));
    const float4 ga = reinterpret_cast<const float4 *>(avg)[((long long)b * 2 * C + c) * plane + i];
    const float4 ba = reinterpret_cast<const float4 *>(avg)[((long long)b * 2 * C + C + c) * plane + i];
    g = make_float4(ace_blend(ag, ga.x, gs.x), ace_blend(ag, ga.y, gs.y), ace_blend(ag, ga.z, gs.z), ace_blend(ag, ga.w, gs.w));
    bt = make_float4(ace_blend(ab, ba.x, bs.x), ace_blend(ab, ba.y, bs.y), ace_blend(ab, ba.z, bs.z), ace_blend(ab, ba.w, bs.w));
  }
  float4 o;
  o.x = ace_tail(xv.x, rv.x, nv, sc, sh, g.x, bt.x, slope);
  o.y = ace_tail(xv.y, rv.y, nv, sc, sh, g.y, bt.y, slope);
  o.z = ace_tail(xv.z, rv.z, nv, sc, sh, g.z, bt.z, slope);
  o.w = ace_tail(xv.w, rv.w, nv, sc, sh, g.w, bt.w, slope);
  reinterpret_cast<float4 *>(out)[((long long)b * C + c) * plane + i] = o;
}

extern "C" int hf_ace_modulate_f32(float *out, const float *x, const float *noise, const float *noise_var,
                                   const float *bn_scale, const float *bn_shift, const float *avg, const float *sp,
                                   const float *blend, int batch, int channels, int hw, int group, float slope, int x_upsample_w,
                                   void *stream) {
  if (!out || !x || !bn_scale || !bn_shift || !sp || batch <= 0 || channels <= 0 || hw <= 0 || (hw & 3) || group <= 0 ||
      (avg && !blend) || batch > 65535 || channels > 65535 || x_upsample_w < 0 ||
      (x_upsample_w && ((x_upsample_w & 3) || hw % x_upsample_w || ((hw / x_upsample_w) & 1))))
    return HF_E_INVALID;
  hipLaunchKernelGGL(ace_modulate, dim3(hf_cdiv(hw / 4, 256), channels, batch), dim3(256), 0, (hipStream_t)stream, out, x, noise,
                     noise_var, bn_scale, bn_shift, avg, sp, blend, channels, hw / 4, group, slope, x_upsample_w, x_upsample_w ? 1 : 0);
  return hf_launch_status();
}

// The same tail with ACE's `avg` planes (conv_gamma / conv_beta of the per-region style map) formed IN the kernel from
// their lookup table instead of read from a [B, 2C, H, W] tensor: avg[b, c, p] = avg_bias[c] + the label_conv3x3 of
// table rows c (gamma) / C + c (beta), columns b*labels.. (see label_conv3x3_v4 - same LDS tables, same interior /
// general arithmetic per pixel, so the result equals hf_label_conv3x3_f32 followed by hf_ace_modulate_f32 bit for bit).
// Saves writing and re-reading 2C planes per sample (at 128^2 x 256 channels x 8 samples: 2 x 268 MB).
// Block = 1024 pixels (four per thread) x `cchunk` channels of one sample.
__global__ __launch_bounds__(256) void ace_modulate_table(float *__restrict__ out, const float *__restrict__ x,
                                                          const float *__restrict__ r, const float *__restrict__ noise_var,
                                                          const float *__restrict__ bn_scale, const float *__restrict__ bn_shift,
                                                          const int *__restrict__ labels, const float *__restrict__ table,
                                                          const float *__restrict__ avg_bias, const float *__restrict__ sp,
                                                          const float *__restrict__ blend, int C, int H, int W, int tcols, int nl,
                                                          int group, float slope, int cchunk, int interior, int x_up) {
  HF_DYN_LDS;
  float *tl = reinterpret_cast<float *>(hf_dyn_lds);  // [9][2*nc][nl]: local row 2*i = gamma of channel c0+i, 2*i+1 = beta
  const int b = blockIdx.z, c0 = blockIdx.y * cchunk, nc = min(cchunk, C - c0);
  float *ts = tl + 9 * 2 * cchunk * nl;
  stage_label_tables(tl, ts, table, 2 * C, tcols, b * nl, nl, 2 * nc, [&](int i) { return (i & 1) * C + c0 + (i >> 1); });
  const int hw = H * W;
  const int p = (blockIdx.x * 256 + threadIdx.x) * 4;
  const bool live = p < hw;
  const int pc = live ? p : 0;
  const int y = pc / W, xx = pc - y * W;
  int lab[3][6];
  label_window(lab, labels + (long long)(b / group) * hw, y, xx, H, W, nl);
  bool inner[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) inner[j] = interior && label_interior(lab, j);
  const bool any_general = hf_wave_any(live && !(inner[0] && inner[1] && inner[2] && inner[3]));
  const float ag = 1.0f / (1.0f + expf(-blend[0])), ab = 1.0f / (1.0f + expf(-blend[1]));
  float4 rv = make_float4(0, 0, 0, 0);
  if (r) rv = *reinterpret_cast<const float4 *>(r + (long long)b * hw + pc);
  const float rr[4] = {rv.x, rv.y, rv.z, rv.w};
  const long long sb = (long long)(b / group) * 2 * C;
  for (int i = 0; i < nc; ++i) {
    const int c = c0 + i;
    float ga[4], ba[4];
    label_lookup4(ga, tl, ts, 2 * nc, nl, 2 * i, avg_bias ? avg_bias[c] : 0.0f, lab, inner, any_general);
    label_lookup4(ba, tl, ts, 2 * nc, nl, 2 * i + 1, avg_bias ? avg_bias[C + c] : 0.0f, lab, inner, any_general);
    float4 xv;
    if (x_up) {  // x [B, C, H/2, W/2]: the nearest-neighbour x2 up-sampling read in place
      const float2 lo = *reinterpret_cast<const float2 *>(x + ((long long)b * C + c) * (hw >> 2) + (long long)(y >> 1) * (W >> 1) + (xx >> 1));
      xv = make_float4(lo.x, lo.x, lo.y, lo.y);
    } else {
      xv = *reinterpret_cast<const float4 *>(x + ((long long)b * C + c) * hw + pc);
    }
    const float4 gs4 = *reinterpret_cast<const float4 *>(sp + (sb + c) * hw + pc);
    const float4 bs4 = *reinterpret_cast<const float4 *>(sp + (sb + C + c) * hw + pc);
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gs[4] = {gs4.x, gs4.y, gs4.z, gs4.w}, bs[4] = {bs4.x, bs4.y, bs4.z, bs4.w};
    const float nv = noise_var ? noise_var[c] : 0.0f, sc = bn_scale[c], sh = bn_shift[c];
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o[j] = ace_tail(xs[j], rr[j], nv, sc, sh, ace_blend(ag, ga[j], gs[j]), ace_blend(ab, ba[j], bs[j]), slope);
    }
    if (live) *reinterpret_cast<float4 *>(out + ((long long)b * C + c) * hw + pc) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

extern "C" int hf_ace_modulate_table_f32(float *out, const float *x, const float *noise, const float *noise_var,
                                         const float *bn_scale, const float *bn_shift, const int *labels, const float *table,
                                         const float *avg_bias, const float *sp, const float *blend, int batch, int channels,
                                         int h, int w, int table_cols, int n_labels, int group, float slope, int interior,
                                         int x_upsample, void *stream) {
  if (!out || !x || !bn_scale || !bn_shift || !labels || !table || !sp || !blend || batch <= 0 || channels <= 0 || h <= 0 ||
      w <= 0 || (w & 3) || group <= 0 || batch > 65535 || n_labels <= 0 || n_labels > kLabelMax ||
      (long long)batch * n_labels > table_cols || (x_upsample && (h & 1)))
    return HF_E_INVALID;
  const int hw = h * w;
  int cchunk = min(channels, 32);
  while (cchunk > 4 && (long long)hf_cdiv(hw, 1024) * hf_cdiv(channels, cchunk) * batch < 1024) cchunk = (cchunk + 1) / 2;
  const size_t lds = (size_t)10 * 2 * cchunk * n_labels * sizeof(float);
  hipLaunchKernelGGL(ace_modulate_table, dim3(hf_cdiv(hw, 1024), hf_cdiv(channels, cchunk), batch), dim3(256), lds,
                     (hipStream_t)stream, out, x, noise, noise_var, bn_scale, bn_shift, labels, table, avg_bias, sp, blend, channels,
                     h, w, table_cols, n_labels, group, slope, cchunk, interior ? 1 : 0, x_upsample ? 1 : 0);
  return hf_launch_status();
}

// Zencoder's region pooling (architecture.py:187-205): out[b, l, c] = mean of act(x[b, c, p]) over the pixels p whose
// label is l, 0 when the label does not occur.  One block per (c, b); every thread keeps the 19 partial sums of its
// pixels in registers (compile-time label loop: no dynamic register indexing), waves reduce by shuffles, the four wave
// results are added in wave order: deterministic.  x rows are `pitch` floats apart, planes `plane_stride` (a view into
// a larger tensor: the conv output computed on the reflection-padded plane).  act: 0 none, 1 tanh.
__global__ __launch_bounds__(256) void region_mean(float *__restrict__ out, const float *__restrict__ x,
                                                   const int *__restrict__ labels, int C, int H, int W, long long batch_stride,
                                                   long long plane_stride, int pitch, int act) {
  HF_DYN_LDS;
  float *red = reinterpret_cast<float *>(hf_dyn_lds);  // [4 waves][2 * kSeanLabels]
  const int c = blockIdx.x, b = blockIdx.y;
  const float *xp = x + (long long)b * batch_stride + (long long)c * plane_stride;
  const int *lb = labels + (long long)b * H * W;
  float s[kSeanLabels], n[kSeanLabels];
#pragma unroll
  for (int l = 0; l < kSeanLabels; ++l) s[l] = n[l] = 0.0f;
  for (int i = threadIdx.x; i < H * W; i += 256) {
    const int yy = i / W, xx = i - yy * W;
    float v = xp[(long long)yy * pitch + xx];
    if (act == 1) v = tanhf(v);
    const int lab = lb[i];
#pragma unroll
    for (int l = 0; l < kSeanLabels; ++l) {
      s[l] += lab == l ? v : 0.0f;
      n[l] += lab == l ? 1.0f : 0.0f;
    }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int l = 0; l < kSeanLabels; ++l) {
    const float sv = hf_wave_sum(s[l]), nv = hf_wave_sum(n[l]);
    if (lane == 0) {
      red[wave * 2 * kSeanLabels + l] = sv;
      red[wave * 2 * kSeanLabels + kSeanLabels + l] = nv;
    }
  }
  __syncthreads();
  if (threadIdx.x < kSeanLabels) {
    const int l = threadIdx.x, st = 2 * kSeanLabels;
    const float tot = (red[l] + red[st + l]) + (red[2 * st + l] + red[3 * st + l]);
    const float cnt = (red[kSeanLabels + l] + red[st + kSeanLabels + l]) + (red[2 * st + kSeanLabels + l] + red[3 * st + kSeanLabels + l]);
    out[((long long)b * kSeanLabels + l) * C + c] = cnt > 0.0f ? tot / cnt : 0.0f;
  }
}

extern "C" int hf_region_mean_f32(float *out, const float *x, const int *labels, int batch, int channels, int h, int w,
                                  int n_labels, long long batch_stride, long long plane_stride, int pitch, int act, void *stream) {
  if (!out || !x || !labels || batch <= 0 || channels <= 0 || h <= 0 || w <= 0 || n_labels != kSeanLabels || pitch < w ||
      batch > 65535 || (act != 0 && act != 1))
    return HF_E_INVALID;
  hipLaunchKernelGGL(region_mean, dim3(channels, batch), dim3(256), 4 * 2 * kSeanLabels * sizeof(float), (hipStream_t)stream, out,
                     x, labels, channels, h, w, batch_stride, plane_stride, pitch, act);
  return hf_launch_status();
}

__global__ __launch_bounds__(256) void tanh_kernel(float *__restrict__ out, const float *__restrict__ x, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = tanhf(x[i]);
}

extern "C" int hf_tanh_f32(float *out, const float *x, long long n, void *stream) {
  if (!out || !x || n <= 0) return HF_E_INVALID;
  hipLaunchKernelGGL(tanh_kernel, dim3(hf_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, out, x, n);
  return hf_launch_status();
}
