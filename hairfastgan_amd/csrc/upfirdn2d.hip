// upfirdn2d.hip - FIR resampling kernels.
//
// hf_upfirdn2d_f32            : general upfirdn2d (any up/down/pad, taps <= 8x8);
//                               reference op/upfirdn2d.py:145-200, upfirdn2d_kernel.cu:49-207.
// hf_blur_noise_bias_act_f32  : the generator's hot instance (mode 1: up=down=1, 4x4,
//                               pad (1,1); model.py:77-93 called at :263) fused with
//                               NoiseInjection + FusedLeakyReLU (model.py:288-293, :341).
//
// Both are HBM-bound streaming kernels: algorithmic traffic is one read of the
// input plane and one write of the output plane (8 B per output element, plus
// the (2H+1)^2 vs (2H)^2 rim).  Lanes map to consecutive output columns so all
// global accesses are coalesced rows; the 4-row sliding window lives in
// registers, so each input element is fetched from L1/L2 at most 4x by
// neighbouring lanes and from HBM once.
#define HF_WANT_F16_SPLIT
#include "hf_common.h"

// Floating-point contraction: "on" = a multiply and an add are fused only where they are written in ONE expression (or as
// fmaf), never across statements.  hipcc's default ("fast") lets the backend fuse opportunistically per basic block: the
// unrolled body of a grid-stride loop and its remainder iterations then round differently, i.e. a sample's bits depend on
// how many elements the launch has - on what it is batched with (found by tools/probes/batch_variance.py in
// upsample_bilinear_add: the third of three images differed from the third of six by one ulp).
#pragma clang fp contract(on)

namespace {

constexpr int kMaxTaps = 8;

__global__ __launch_bounds__(256) void upfirdn2d_generic(
    float *__restrict__ out, const float *__restrict__ in, const float *__restrict__ kernel, int in_h,
    int in_w, int out_h, int out_w, int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0,
    int pad_y0, long long total) {
  HF_DYN_LDS;
  float *kf = reinterpret_cast<float *>(hf_dyn_lds);  // flipped taps: kf[ky][kx] = k[kh-1-ky][kw-1-kx]
  for (int i = threadIdx.x; i < kh * kw; i += blockDim.x) {
    int ky = i / kw, kx = i - ky * kw;
    kf[i] = kernel[(kh - 1 - ky) * kw + (kw - 1 - kx)];
  }
  __syncthreads();
  long long stride = (long long)gridDim.x * blockDim.x;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
    int ox = (int)(idx % out_w);
    long long t = idx / out_w;
    int oy = (int)(t % out_h);
    long long plane = t / out_h;
    const float *src = in + plane * (long long)in_h * in_w;
    float acc = 0.0f;
    for (int ky = 0; ky < kh; ++ky) {
      int uy = oy * down_y + ky - pad_y0;  // row in the zero-inserted signal
      if (uy < 0 || uy % up_y) continue;
      int iy = uy / up_y;
      if (iy >= in_h) continue;
      for (int kx = 0; kx < kw; ++kx) {
        int ux = ox * down_x + kx - pad_x0;
        if (ux < 0 || ux % up_x) continue;
        int ix = ux / up_x;
        if (ix >= in_w) continue;
        acc = fmaf(src[(long long)iy * in_w + ix], kf[ky * kw + kx], acc);
      }
    }
    out[idx] = acc;
  }
}

// Blur 4x4, pad (1,1): out[oy][ox] = sum_{ky,kx} kf[ky][kx] * in[oy+ky-1][ox+kx-1].
// blockDim = (64 columns, 4 row segments); each thread walks kRowsPerThread
// output rows keeping the 4x4 input window in registers.
constexpr int kBlurCols = 64;
constexpr int kBlurSegs = 4;
constexpr int kRowsPerThread = 32;

__global__ __launch_bounds__(256) void blur4x4_noise_bias_act(
    float *__restrict__ out, const float *__restrict__ in, const float *__restrict__ kernel4x4,
    const float *__restrict__ noise, const float *__restrict__ noise_w, long long noise_bstride,
    const float *__restrict__ bias, int channels, int in_h, int in_w, int in_pitch, float alpha, float scale) {
  const int out_h = in_h - 1, out_w = in_w - 1;
  const int ox = blockIdx.x * kBlurCols + threadIdx.x;
  const int oy0 = (blockIdx.y * kBlurSegs + threadIdx.y) * kRowsPerThread;
  const int plane = blockIdx.z;
  if (ox >= out_w || oy0 >= out_h) return;

  float kf[4][4];
#pragma unroll
  for (int ky = 0; ky < 4; ++ky)
#pragma unroll
    for (int kx = 0; kx < 4; ++kx) kf[ky][kx] = kernel4x4[(3 - ky) * 4 + (3 - kx)];

  const float *src = in + (long long)plane * in_h * in_pitch;
  float *dst = out + (long long)plane * out_h * out_w;
  const int c = plane % channels;
  const int b = plane / channels;
  const float bc = bias ? bias[c] : 0.0f;
  const float nw = noise ? noise_w[0] : 0.0f;
  const float *nz = noise ? noise + (long long)b * noise_bstride : nullptr;

  // column validity of the 4 taps (input columns ox-1 .. ox+2)
  bool cv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) cv[j] = (ox - 1 + j) >= 0 && (ox - 1 + j) < in_w;

  // Sliding window over input rows, 4 output rows per step: the 16 loads of a step are
  // independent and issued together (4x the bytes in flight of a row-at-a-time walk,
  // which was latency bound at ~2.2 TB/s).
  float win[7][4];  // win[r][j]: input row (oy-1+r), column (ox-1+j)
  auto load_row = [&](int iy, float (&row)[4]) {
    const bool rv = iy >= 0 && iy < in_h;
    const float *p = src + (long long)iy * in_pitch + (ox - 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) row[j] = (rv && cv[j]) ? p[j] : 0.0f;
  };
#pragma unroll
  for (int r = 0; r < 3; ++r) load_row(oy0 - 1 + r, win[r]);

  const int oy_end = min(oy0 + kRowsPerThread, out_h);
  for (int oy = oy0; oy < oy_end; oy += 4) {
#pragma unroll
    for (int r = 0; r < 4; ++r) load_row(oy + 2 + r, win[3 + r]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (oy + q < oy_end) {
        float acc = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc = fmaf(win[q + r][j], kf[r][j], acc);
        if (nz) acc = fmaf(nw, nz[(long long)(oy + q) * out_w + ox], acc);
        if (bias) acc = hf_lrelu(acc + bc, alpha, scale);
        dst[(long long)(oy + q) * out_w + ox] = acc;
      }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) win[r][j] = win[r + 4][j];
  }
}


// Vector form for out_w % 4 == 0 (every generator layer): a thread owns 4 consecutive
// output columns c0..c0+3 and needs input columns c0-1..c0+5 of each row.  It loads
// columns c0..c0+3 with one unaligned 16-byte load (input rows have odd length 2W+1);
// the three edge columns come from the neighbouring lanes' registers (wave shuffles),
// only the first / last lane of a wave touch memory for them.  Two output rows per
// iteration keep two row loads in flight; one aligned 16-byte store per output row.
// blockDim = (cq column quads, 256/cq row segments), cq = min(64, out_w/4).
struct __attribute__((packed, aligned(4))) f32x4u {
  float x, y, z, w;
};

__global__ __launch_bounds__(256) void blur4x4_noise_bias_act_vec4(
    float *__restrict__ out, const float *__restrict__ in, const float *__restrict__ kernel4x4,
    const float *__restrict__ noise, const float *__restrict__ noise_w, long long noise_bstride,
    const float *__restrict__ bias, int channels, int in_h, int in_w, int in_pitch, float alpha, float scale,
    int rows_per_thread) {
  const int out_h = in_h - 1, out_w = in_w - 1;
  const int c0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int oy0 = (blockIdx.y * blockDim.y + threadIdx.y) * rows_per_thread;
  const int plane = blockIdx.z;
  // no early return: every lane takes part in the shuffles; inactive ones just do not store
  const bool active = c0 < out_w && oy0 < out_h;
  const int lane = (threadIdx.y * blockDim.x + threadIdx.x) & 63;
  // neighbours within the wave AND within the same row segment hold columns c0-4.. / c0+4..
  const bool nb_left = lane > 0 && threadIdx.x > 0;
  const bool nb_right = lane < 63 && threadIdx.x + 1 < blockDim.x && c0 + 4 < out_w;  // neighbour must be active

  float kf[4][4];
#pragma unroll
  for (int ky = 0; ky < 4; ++ky)
#pragma unroll
    for (int kx = 0; kx < 4; ++kx) kf[ky][kx] = kernel4x4[(3 - ky) * 4 + (3 - kx)];

  const float *src = in + (long long)plane * in_h * in_pitch;
  float *dst = out + (long long)plane * out_h * out_w;
  const int c = plane % channels;
  const int b = plane / channels;
  const float bc = bias ? bias[c] : 0.0f;
  const float nw = noise ? noise_w[0] : 0.0f;
  const float *nz = noise ? noise + (long long)b * noise_bstride : nullptr;

  float win[6][7];  // win[r][j]: input row oy-1+r, column c0-1+j
  // A row is fetched in two steps so that several rows' loads are in flight together: `issue`
  // starts the 16-byte load (+ the two edge dwords on the first/last lane of a segment),
  // `finish` does the neighbour shuffles, which need the loaded data.
  struct RowRaw {
    f32x4u m;
    float e0, e5, e6;
  };
  auto issue = [&](int iy) {
    RowRaw r;
    const bool rv = active && iy >= 0 && iy < in_h;
    const float *p = src + (long long)iy * in_pitch + c0;
    r.m = f32x4u{0.f, 0.f, 0.f, 0.f};
    if (rv) r.m = *reinterpret_cast<const f32x4u *>(p);
    r.e0 = (!nb_left && rv && c0 > 0) ? p[-1] : 0.0f;
    r.e5 = (!nb_right && rv && c0 + 4 < in_w) ? p[4] : 0.0f;
    r.e6 = (!nb_right && rv && c0 + 5 < in_w) ? p[5] : 0.0f;
    return r;
  };
  auto finish = [&](const RowRaw &r, float (&row)[7]) {
    row[1] = r.m.x; row[2] = r.m.y; row[3] = r.m.z; row[4] = r.m.w;
    const float l = __shfl_up(r.m.w, 1, 64), r0 = __shfl_down(r.m.x, 1, 64), r1 = __shfl_down(r.m.y, 1, 64);
    row[0] = nb_left ? l : r.e0;
    row[5] = nb_right ? r0 : r.e5;
    row[6] = nb_right ? r1 : r.e6;
  };
  {
    RowRaw a = issue(oy0 - 1), b2 = issue(oy0), c2 = issue(oy0 + 1), d2 = issue(oy0 + 2);
    finish(a, win[0]); finish(b2, win[1]); finish(c2, win[2]); finish(d2, win[3]);
  }

  const int oy_end = min(oy0 + rows_per_thread, out_h);
  RowRaw nx0 = issue(oy0 + 3), nx1 = issue(oy0 + 4);  // rows for the first iteration
  for (int oy = oy0; oy < oy0 + rows_per_thread; oy += 2) {  // uniform trip count (shuffles inside)
    finish(nx0, win[4]);
    finish(nx1, win[5]);
    nx0 = issue(oy + 5);  // next iteration's rows: in flight during this iteration's FMAs / stores
    nx1 = issue(oy + 6);
#pragma unroll
    for (int q2 = 0; q2 < 2; ++q2) {
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q] = fmaf(win[q2 + r][q + j], kf[r][j], acc[q]);
      const int oyy = oy + q2;
      if (active && oyy < oy_end) {
        if (nz) {
          const float4 z = *reinterpret_cast<const float4 *>(nz + (long long)oyy * out_w + c0);
          acc[0] = fmaf(nw, z.x, acc[0]); acc[1] = fmaf(nw, z.y, acc[1]);
          acc[2] = fmaf(nw, z.z, acc[2]); acc[3] = fmaf(nw, z.w, acc[3]);
        }
        if (bias) {
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q] = hf_lrelu(acc[q] + bc, alpha, scale);
        }
        *reinterpret_cast<float4 *>(dst + (long long)oyy * out_w + c0) =
            make_float4(acc[0], acc[1], acc[2], acc[3]);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < 7; ++j) win[r][j] = win[r + 2][j];
  }
}

}  // namespace

extern "C" int hf_upfirdn2d_f32(float *out, const float *in, const float *kernel, int major, int in_h,
                                int in_w, int kh, int kw, int up_x, int up_y, int down_x, int down_y,
                                int pad_x0, int pad_x1, int pad_y0, int pad_y1, void *stream) {
  if (!out || !in || !kernel || major <= 0 || in_h <= 0 || in_w <= 0 || kh <= 0 || kw <= 0 ||
      kh > kMaxTaps || kw > kMaxTaps || up_x <= 0 || up_y <= 0 || down_x <= 0 || down_y <= 0)
    return HF_E_INVALID;
  const int full_h = in_h * up_y + pad_y0 + pad_y1 - kh;
  const int full_w = in_w * up_x + pad_x0 + pad_x1 - kw;
  if (full_h < 0 || full_w < 0) return HF_E_INVALID;
  const int out_h = full_h / down_y + 1, out_w = full_w / down_x + 1;
  const long long total = (long long)major * out_h * out_w;
  long long g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(upfirdn2d_generic, dim3((int)g), dim3(256), kMaxTaps * kMaxTaps * sizeof(float),
                     (hipStream_t)stream, out, in, kernel, in_h, in_w, out_h, out_w, kh, kw, up_x, up_y,
                     down_x, down_y, pad_x0, pad_y0, total);
  return hf_launch_status();
}

// ------------------------------------------------------------------------------------------
// Blur + noise + bias + lrelu whose result feeds a 3x3 conv on the fp16 matrix cores directly:
// instead of the fp32 NCHW activation it writes s_next[b,c] * y already SPLIT into fp16 (hi, lo)
// pairs and K-BLOCKED, hi/lo[b][c/8][y][x][8 halves] - the exact units csrc/convh.hip stages into
// LDS, so that kernel fetches them by LDS-DMA with no per-element loads or conversion.  Values
// are bit-identical to what convh's own staging derives from the fp32 activation (same tap
// order, fp32 product with s, hi = fp16(v), lo = fp16(v - hi)).
// A lane owns one INPUT column (x-1 of its output column) and 8 channels and walks down the rows:
// per input row and channel one coalesced dword load; the three further columns of the 4-tap
// window come from the next lanes' registers (wave shuffles), so a wave of 64 input columns
// produces 61 output columns and every input byte is requested once.  Four rotating accumulators
// per channel hold the output rows an input row contributes to.  Writes are 16 B per lane for hi
// and for lo (up to 1 KiB contiguous per wave).
typedef _Float16 hf_half8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512) void blur4x4_split8(hf_half8 *__restrict__ hi, hf_half8 *__restrict__ lo,
                                                      const float *__restrict__ in,
                                                      const float *__restrict__ kernel4x4,
                                                      const float *__restrict__ noise,
                                                      const float *__restrict__ noise_w, long long noise_bstride,
                                                      const float *__restrict__ bias,
                                                      const float *__restrict__ s_next, int channels, int in_h,
                                                      int in_w, int in_pitch, float alpha, float scale,
                                                      int rows_per_thread) {
  const int out_h = in_h - 1, out_w = in_w - 1;
  // waves tile the columns with a stride of 61: lane l of wave wv holds input column
  // ix = 61*wv + l - 1 and produces output column ox = ix + 1 (lanes 61..63 only feed neighbours)
  const int lane = threadIdx.x & 63;
  const int wv = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int ix = 61 * wv + lane - 1;
  const int ox = ix + 1;
  const int oy0 = blockIdx.y * rows_per_thread;
  const int cblocks = channels >> 3;
  const int b = blockIdx.z / cblocks, cb = blockIdx.z - b * cblocks;
  if (61 * wv >= out_w || oy0 >= out_h) return;  // whole wave out of range (uniform)
  const bool produces = lane < 61 && ox < out_w;

  float kf[4][4];
#pragma unroll
  for (int ky = 0; ky < 4; ++ky)
#pragma unroll
    for (int kx = 0; kx < 4; ++kx) kf[ky][kx] = kernel4x4[(3 - ky) * 4 + (3 - kx)];
  float bc[8], sv[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    bc[k] = bias ? bias[cb * 8 + k] : 0.0f;
    sv[k] = s_next ? s_next[(long long)b * channels + cb * 8 + k] : 1.0f;
  }
  const float nw = noise ? noise_w[0] : 0.0f;
  const float *nz = noise ? noise + (long long)b * noise_bstride : nullptr;
  const long long iplane = (long long)in_h * in_pitch;
  const bool colv = ix >= 0 && ix < in_w;
  const float *src = in + ((long long)b * channels + cb * 8) * iplane + (colv ? ix : 0);
  const int oxc = min(ox, out_w - 1);
  hf_half8 *hp = hi + ((long long)b * cblocks + cb) * out_h * out_w + oxc;
  hf_half8 *lp = lo + ((long long)b * cblocks + cb) * out_h * out_w + oxc;

  const int oy_end = min(oy0 + rows_per_thread, out_h);
  float acc[8][4];  // acc[k][slot]: output row oy with (oy - oy0) % 4 == slot, channel k
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[k][t] = 0.0f;

  // input rows iy = oy0 - 1 + n, n = 0 .. (oy_end - oy0) + 2, four per iteration: the 32 row loads
  // of an iteration are issued before the first use, and the slots of the rotating accumulators
  // are compile-time constants
  const int n_end = (oy_end - oy0) + 3;
  for (int n0 = 0; n0 < n_end; n0 += 4) {
    float own[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int iy = oy0 - 1 + n0 + u;
      const bool rv = colv && iy >= 0 && iy < in_h && (n0 + u) < n_end;
      const long long ro = (long long)min(max(iy, 0), in_h - 1) * in_pitch;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float v = src[k * iplane + ro];  // unconditional, clamped address
        own[u][k] = rv ? v : 0.0f;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int n = n0 + u;
      if (n < n_end) {  // uniform
        // kernel row r of output row oy = oy0 + n - r (r ascending per output row, columns inner:
        // the accumulation order of blur4x4_noise_bias_act); window = own column and the next three
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float win[4];
          // the next three lanes' columns by DPP wavefront shifts (one VALU instruction each; __shfl_down is a ds_bpermute
          // with an address register: 96 LDS round trips per four input rows).  Lanes 61-63, whose shifted values run off the
          // wave, only feed their neighbours
          win[0] = own[u][k];
          win[1] = hf_lane_down(win[0]);
          win[2] = hf_lane_down(win[1]);
          win[3] = hf_lane_down(win[2]);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int slot = (u - r) & 3;
            if (n - r >= 0) {
              float a = (r == 0) ? 0.0f : acc[k][slot];
#pragma unroll
              for (int j = 0; j < 4; ++j) a = fmaf(win[j], kf[r][j], a);
              acc[k][slot] = a;
            }
          }
        }
        // the output row whose last kernel row just arrived
        const int oy = oy0 + n - 3;
        if (n >= 3 && oy < oy_end && produces) {
          const int slot = (u - 3) & 3;
          const float nzr = nz ? nz[(long long)oy * out_w + ox] : 0.0f;
          hf_half8 h8, l8;
          bool ovf = false;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            float v = acc[k][slot];
            if (nz) v = fmaf(nw, nzr, v);
            if (bias) v = hf_lrelu(v + bc[k], alpha, scale);
            v *= sv[k];
            _Float16 hv, lv;
            hf_split_f16(v, hv, lv, ovf);  // hi and lo from the same fp32-rounded product, saturating (hf_common.h)
            h8[k] = hv;
            l8[k] = lv;
          }
          hf_note_overflow(ovf);
          hp[(long long)oy * out_w] = h8;
          if (lo) lp[(long long)oy * out_w] = l8;  // lo == nullptr: plain fp16 consumer (nterms 1)
        }
      }
    }
  }
}

extern "C" unsigned long long hf_f16_overflow_count_blur(int reset) { return hf_f16_overflow_read_tu(reset); }

extern "C" int hf_blur_noise_bias_act_f32(float *out, const float *in, const float *kernel4x4,
                                          const float *noise, const float *noise_w,
                                          long long noise_bstride, const float *bias, int batch,
                                          int channels, int in_h, int in_w, int in_pitch, float alpha,
                                          float scale, void *stream) {
  if (!out || !in || !kernel4x4 || batch <= 0 || channels <= 0 || in_h < 2 || in_w < 2 || in_pitch < in_w ||
      (noise && !noise_w))
    return HF_E_INVALID;
  const long long planes = (long long)batch * channels;
  if (planes > 65535LL * 32768) return HF_E_INVALID;
  const int out_h = in_h - 1, out_w = in_w - 1;
  const bool aligned = ((((size_t)out) | ((size_t)noise)) & 15) == 0 && (noise_bstride & 3) == 0;
  if ((out_w & 3) == 0 && aligned) {
    int cq = 64;  // column quads per block row (power of two, <= out_w / 4 rounded up)
    while (cq > 1 && cq * 4 >= 2 * out_w) cq >>= 1;
    const int segs = 256 / cq;
    int rpt = 32;  // rows per thread: even, shrink for short planes so that segments are not idle
    while (rpt > 2 && segs * rpt >= 2 * out_h) rpt >>= 1;
    dim3 grid(hf_cdiv(out_w, cq * 4), hf_cdiv(out_h, segs * rpt), (unsigned)planes);
    hipLaunchKernelGGL(blur4x4_noise_bias_act_vec4, grid, dim3(cq, segs), 0, (hipStream_t)stream, out, in,
                       kernel4x4, noise, noise_w, noise_bstride, bias, channels, in_h, in_w, in_pitch, alpha, scale, rpt);
    return hf_launch_status();
  }
  dim3 grid(hf_cdiv(out_w, kBlurCols), hf_cdiv(out_h, kBlurSegs * kRowsPerThread), (unsigned)planes);
  hipLaunchKernelGGL(blur4x4_noise_bias_act, grid, dim3(kBlurCols, kBlurSegs), 0, (hipStream_t)stream, out,
                     in, kernel4x4, noise, noise_w, noise_bstride, bias, channels, in_h, in_w, in_pitch, alpha, scale);
  return hf_launch_status();
}

extern "C" int hf_blur_noise_bias_act_split_f16(void *out_hi, void *out_lo, const float *in, const float *kernel4x4,
                                                const float *noise, const float *noise_w, long long noise_bstride,
                                                const float *bias, const float *s_next, int batch, int channels,
                                                int in_h, int in_w, int in_pitch, float alpha, float scale,
                                                void *stream) {
  if (!out_hi || !in || !kernel4x4 || batch <= 0 || channels <= 0 || (channels & 7) || in_h < 2 || in_w < 2 ||
      in_pitch < in_w || (noise && !noise_w))
    return HF_E_INVALID;
  const long long zs = (long long)batch * (channels >> 3);
  if (zs > 65535) return HF_E_INVALID;
  const int out_h = in_h - 1, out_w = in_w - 1;
  const int waves = hf_cdiv(out_w, 61);  // 61 output columns per wave
  int wpb = 1, waste = 1 << 30;  // waves per block (<= 8) leaving the fewest idle wave slots
  for (int k = 8; k >= 1; --k) {
    const int ws = hf_cdiv(waves, k) * k - waves;
    if (ws < waste) { waste = ws; wpb = k; }
  }
  const int tx = 64 * wpb;
  int rpt = 64;  // rows per thread: 3 warm-up rows per strip
  while (rpt > 8 && (long long)hf_cdiv(waves, wpb) * hf_cdiv(out_h, rpt) * zs * wpb < 8192) rpt >>= 1;
  dim3 grid(hf_cdiv(waves, wpb), hf_cdiv(out_h, rpt), (unsigned)zs);
  hipLaunchKernelGGL(blur4x4_split8, grid, dim3(tx), 0, (hipStream_t)stream, static_cast<hf_half8 *>(out_hi),
                     static_cast<hf_half8 *>(out_lo), in, kernel4x4, noise, noise_w, noise_bstride, bias, s_next, channels,
                     in_h, in_w, in_pitch, alpha, scale, rpt);
  return hf_launch_status();
}
