// torgb.hip - ToRGB: 1x1 modulated conv (no demodulation) + bias + upsampled skip.
// Reference: models/stylegan2/model.py:356-365 (ToRGB.forward), :49-53 (Upsample,
// upfirdn2d up=2 pad=(2,1), i.e. upfirdn2d_kernel.cu mode 3).
//
// HBM-bound: per image it streams the whole feature map once (cin*H*W*4 B) and
// writes 3*H*W*4 B; the 3 x cin modulated weights sit in LDS.  Each thread owns
// VEC consecutive pixels (16 B accesses for VEC=4), lanes cover consecutive
// pixel groups so every channel-plane read is a coalesced row segment; the
// channel loop is unrolled 8x to keep 8 independent 16 B loads in flight.
#include "hf_common.h"

// Floating-point contraction: "on" = a multiply and an add are fused only where they are written in ONE expression (or as
// fmaf), never across statements.  hipcc's default ("fast") lets the backend fuse opportunistically per basic block: the
// unrolled body of a grid-stride loop and its remainder iterations then round differently, i.e. a sample's bits depend on
// how many elements the launch has - on what it is batched with (found by tools/probes/batch_variance.py in
// upsample_bilinear_add: the third of three images differed from the third of six by one ulp).
#pragma clang fp contract(on)

namespace {

template <int VEC>
__global__ __launch_bounds__(256) void torgb_kernel(float *__restrict__ out, const float *__restrict__ x,
                                                    const float *__restrict__ wt,
                                                    const float *__restrict__ s,
                                                    const float *__restrict__ bias,
                                                    const float *__restrict__ skip,
                                                    const float *__restrict__ kernel4x4, int cin, int h,
                                                    int w) {
  HF_DYN_LDS;
  float *wm = reinterpret_cast<float *>(hf_dyn_lds);  // wm[c*cin + ci] = wt[ci][c] * s[b][ci]
  const int b = blockIdx.y;
  const int hw = h * w;
  for (int i = threadIdx.x; i < cin; i += blockDim.x) {
    float sv = s ? s[(long long)b * cin + i] : 1.0f;
    wm[i] = wt[i * 3 + 0] * sv;
    wm[cin + i] = wt[i * 3 + 1] * sv;
    wm[2 * cin + i] = wt[i * 3 + 2] * sv;
  }
  __syncthreads();

  const int p0 = (blockIdx.x * blockDim.x + threadIdx.x) * VEC;
  if (p0 >= hw) return;
  const float *xb = x + (long long)b * cin * hw + p0;
  float acc[3][VEC];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[c][v] = 0.0f;

#pragma unroll 8
  for (int ci = 0; ci < cin; ++ci) {
    float xv[VEC];
    if constexpr (VEC == 4) {
      float4 t = *reinterpret_cast<const float4 *>(xb + (long long)ci * hw);
      xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
    } else {
      xv[0] = xb[(long long)ci * hw];
    }
    const float w0 = wm[ci], w1 = wm[cin + ci], w2 = wm[2 * cin + ci];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      acc[0][v] = fmaf(w0, xv[v], acc[0][v]);
      acc[1][v] = fmaf(w1, xv[v], acc[1][v]);
      acc[2][v] = fmaf(w2, xv[v], acc[2][v]);
    }
  }

  const int sh = h >> 1, sw = w >> 1;
  if constexpr (VEC == 4) {
    if (skip && (w & 3) == 0) {
      // The four pixels of the thread share a row and start at a multiple of 4: their 2x2 taps of the zero-insert
      // upsampler (pad (2,1), 4x4 true convolution: only taps with the parity of (y, x) hit non-zero samples) read skip
      // rows iy0, iy1 and columns c-1 .. c+2 (c = x0/2): 8 loads per channel instead of 16, the 16 kernel taps once
      // per thread, no per-pixel division.  Same products in the same order as the per-pixel form below.
      float kk[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) kk[i] = kernel4x4[i];
      const int y = p0 / w, x0 = p0 - y * w, cc = x0 >> 1;
      const int ky0 = y & 1, ky1 = ky0 + 2;
      const int iy0 = (y + ky0 - 2) >> 1, iy1 = (y + ky1 - 2) >> 1;
      float kr[2][4];  // the two kernel rows (3 - ky) this output row uses: selected once, indexed by constants below
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) kr[a][j] = ky0 ? kk[(3 - (1 + 2 * a)) * 4 + j] : kk[(3 - 2 * a) * 4 + j];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float bc = bias ? bias[c] : 0.0f;
        const float *sp = skip + ((long long)b * 3 + c) * sh * sw;
        float sv[2][4];  // rows iy0 / iy1, columns cc-1 .. cc+2 (0 outside the plane)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const int iy = a ? iy1 : iy0;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int ix = cc - 1 + j;
            sv[a][j] = (iy >= 0 && iy < sh && ix >= 0 && ix < sw) ? sp[iy * sw + ix] : 0.0f;
          }
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          float r = acc[c][v] + bc;
          float up = 0.0f;
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            const int iy = a ? iy1 : iy0;
            if (iy < 0 || iy >= sh) continue;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const int kx = (v & 1) + 2 * e;            // xx & 1 == v & 1 (x0 is even)
              const int j = ((v + kx - 2) >> 1) + 1;     // compile-time column of sv: ix = (x0 + v + kx - 2) / 2 = cc - 1 + j
              const int ix = cc - 1 + j;
              if (ix < 0 || ix >= sw) continue;
              up = fmaf(sv[a][j], kr[a][3 - kx], up);
            }
          }
          acc[c][v] = r + up;
        }
        float *o = out + ((long long)b * 3 + c) * hw + p0;
        *reinterpret_cast<float4 *>(o) = make_float4(acc[c][0], acc[c][1], acc[c][2], acc[c][3]);
      }
      return;
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float bc = bias ? bias[c] : 0.0f;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      float r = acc[c][v] + bc;
      if (skip) {
        // zero-insert x2, pad (2,1), 4x4 true convolution: only taps with the
        // parity of (y, x) hit non-zero samples -> 2x2 taps per output.
        const int p = p0 + v;
        const int y = p / w, xx = p - y * w;
        const float *sp = skip + ((long long)b * 3 + c) * sh * sw;
        float up = 0.0f;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const int ky = (y & 1) + 2 * a;
          const int iy = (y + ky - 2) >> 1;  // (y+ky-2) is even; >= -1
          if (iy < 0 || iy >= sh) continue;
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int kx = (xx & 1) + 2 * e;
            const int ix = (xx + kx - 2) >> 1;
            if (ix < 0 || ix >= sw) continue;
            up = fmaf(sp[iy * sw + ix], kernel4x4[(3 - ky) * 4 + (3 - kx)], up);
          }
        }
        r += up;
      }
      acc[c][v] = r;
    }
    float *o = out + ((long long)b * 3 + c) * hw + p0;
    if constexpr (VEC == 4) {
      *reinterpret_cast<float4 *>(o) = make_float4(acc[c][0], acc[c][1], acc[c][2], acc[c][3]);
    } else {
      o[0] = acc[c][0];
    }
  }
}

// Small planes (<= 64^2): the streaming kernel above walks all cin channels in one thread - 64
// dependent load batches, ~37 us even for a 4x4 plane.  Here a block is 64 pixels x GROUPS channel
// groups; every group sums its cin/GROUPS channels, the partial sums meet in LDS (fixed order:
// deterministic) and group 0 runs the same epilogue.
template <int GROUPS>
__global__ __launch_bounds__(64 * GROUPS) void torgb_small_kernel(float *__restrict__ out, const float *__restrict__ x,
                                                                   const float *__restrict__ wt,
                                                                   const float *__restrict__ s,
                                                                   const float *__restrict__ bias,
                                                                   const float *__restrict__ skip,
                                                                   const float *__restrict__ kernel4x4, int cin, int h,
                                                                   int w) {
  HF_DYN_LDS;
  float *wm = reinterpret_cast<float *>(hf_dyn_lds);  // wm[c*cin + ci] = wt[ci][c] * s[b][ci]
  float *red = wm + 3 * cin;                           // [GROUPS][3][64]
  const int b = blockIdx.y;
  const int hw = h * w;
  for (int i = threadIdx.x; i < cin; i += blockDim.x) {
    const float sv = s ? s[(long long)b * cin + i] : 1.0f;
    wm[i] = wt[i * 3 + 0] * sv;
    wm[cin + i] = wt[i * 3 + 1] * sv;
    wm[2 * cin + i] = wt[i * 3 + 2] * sv;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int p = blockIdx.x * 64 + lane;
  const bool pv = p < hw;
  const int per = (cin + GROUPS - 1) / GROUPS;
  const int c_lo = grp * per, c_hi = min(cin, c_lo + per);
  const float *xb = x + (long long)b * cin * hw + (pv ? p : 0);
  float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
#pragma unroll 8
  for (int ci = c_lo; ci < c_hi; ++ci) {
    const float xv = xb[(long long)ci * hw];
    a0 = fmaf(wm[ci], xv, a0);
    a1 = fmaf(wm[cin + ci], xv, a1);
    a2 = fmaf(wm[2 * cin + ci], xv, a2);
  }
  red[(grp * 3 + 0) * 64 + lane] = a0;
  red[(grp * 3 + 1) * 64 + lane] = a1;
  red[(grp * 3 + 2) * 64 + lane] = a2;
  __syncthreads();
  if (grp != 0 || !pv) return;
  const int sh = h >> 1, sw = w >> 1;
  const int y = p / w, xx = p - y * w;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float r = 0.0f;
    for (int g = 0; g < GROUPS; ++g) r += red[(g * 3 + c) * 64 + lane];
    r += bias ? bias[c] : 0.0f;
    if (skip) {
      const float *sp = skip + ((long long)b * 3 + c) * sh * sw;
      float up = 0.0f;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int ky = (y & 1) + 2 * a;
        const int iy = (y + ky - 2) >> 1;
        if (iy < 0 || iy >= sh) continue;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int kx = (xx & 1) + 2 * e;
          const int ix = (xx + kx - 2) >> 1;
          if (ix < 0 || ix >= sw) continue;
          up = fmaf(sp[iy * sw + ix], kernel4x4[(3 - ky) * 4 + (3 - kx)], up);
        }
      }
      r += up;
    }
    out[((long long)b * 3 + c) * hw + p] = r;
  }
}

}  // namespace

extern "C" int hf_torgb_f32(float *out, const float *x, const float *wt, const float *s, const float *bias,
                            const float *skip, const float *kernel4x4, int batch, int cin, int h, int w,
                            void *stream) {
  if (!out || !x || !wt || batch <= 0 || batch > 65535 || cin <= 0 || h <= 0 || w <= 0) return HF_E_INVALID;
  if (skip && (!kernel4x4 || (h & 1) || (w & 1))) return HF_E_INVALID;
  const size_t lds = (size_t)3 * cin * sizeof(float);
  if (lds > 64 * 1024) return HF_E_INVALID;
  const int hw = h * w;
  hipStream_t st = (hipStream_t)stream;
  if (hw <= 64 * 64 && cin >= 64) {  // small plane, long channel loop: split the channels over 16 waves
    constexpr int GROUPS = 16;
    dim3 grid(hf_cdiv(hw, 64), batch);
    hipLaunchKernelGGL(torgb_small_kernel<GROUPS>, grid, dim3(64 * GROUPS), lds + GROUPS * 3 * 64 * sizeof(float), st, out, x,
                       wt, s, bias, skip, kernel4x4, cin, h, w);
    return hf_launch_status();
  }
  const bool aligned = ((((size_t)out) | ((size_t)x)) & 15) == 0;
  if (hw % 4 == 0 && aligned) {
    dim3 grid(hf_cdiv(hw / 4, 256), batch);
    hipLaunchKernelGGL(torgb_kernel<4>, grid, dim3(256), lds, st, out, x, wt, s, bias, skip, kernel4x4, cin, h,
                       w);
  } else {
    dim3 grid(hf_cdiv(hw, 256), batch);
    hipLaunchKernelGGL(torgb_kernel<1>, grid, dim3(256), lds, st, out, x, wt, s, bias, skip, kernel4x4, cin, h,
                       w);
  }
  return hf_launch_status();
}
