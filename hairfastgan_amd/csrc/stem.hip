// stem.hip - the ResNet stem of BiSeNet's context path (models/CtrlHair/external_code/face_parsing/resnet.py:57-62,
// 81-84): conv1 (3 -> 64, 7x7, stride 2, pad 3, no bias) + bn1 + ReLU + MaxPool2d(3, 2, 1), on the fp16 matrix cores and
// in ONE pass - the 64-channel half-resolution activation (1 GB for sixteen 1024^2 images) is never written.
//
// The general fp32-MFMA conv ran this layer at 4.3 ms per sixteen 1024^2 images (3 input channels fill 3/8 of a K step,
// one tap at a time) and the pooling pass re-read its output.  Here the layer is a GEMM over K = (ci, ky) groups of
// eight kx taps (seven real, the eighth multiplies a zero weight): 21 groups + 1 zero group = 11 MFMA K-steps of 16.
//   * activations: the block's input window as fp16 (hi, lo) pairs in LDS, [part][ci][row][72 columns]; the B operand of
//     conv pixel (r, c), group (ci, ky) is the eight consecutive halves of row 2r+ky from column 2c on - no im2col copy;
//   * weights: [group][64 channels][8 halves] hi / lo in LDS for the whole kernel (45 KB), A operand = one 16-byte read;
//   * f16x3: hi*hi + hi*lo + lo*hi in the fp32 accumulator (csrc/convh.hip), weights pre-scaled by a power of two.
// POOL: a wave owns five conv rows (4w .. 4w+4 of the tile's 17) = the windows of two pooled rows; the 3x3 maximum is
// taken in registers (columns from the neighbouring lanes), so blocks overlap by one conv row / column instead of
// exchanging them: a block of four waves yields 8 x 15 pooled pixels of 64 channels from 20 x 32 conv pixels.
// Measured (tools/probes/stem.py): sixteen 1024^2 images 4.81 ms (conv + pool kernels) -> 0.82 ms, 24 x 512^2 1.80 -> 0.27 ms.
// What is left is vector-ALU time, not matrix time: with loads, MFMAs and stores all removed the kernel still takes half
// of that - ~50 instructions per staged window element (index arithmetic, saturating split, two 2-byte LDS stores) and
// ~17 per pooled accumulator value in the epilogue.
#define HF_WANT_F16_SPLIT
#include "conv_common.h"

using namespace hf_detail;

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr int kStemGroups = 22;   // K groups of 8: (ci, ky) = group / 7, group % 7 for the first 21; the last is zero
constexpr int kStemSteps = 11;    // MFMA K-steps of 16
constexpr int kStemPitch = 72;    // halves per window row: 2*31 + 7 = 69 used, + the zero-weight eighth tap

template <bool POOL>
struct StemGeom {
  static constexpr int PG = POOL ? 5 : 4;             // conv rows per wave
  static constexpr int ROW_STEP = 4;                  // first conv row of wave w = 4*w
  static constexpr int CONV_ROWS = POOL ? 17 : 16;    // per tile
  static constexpr int WIN_ROWS = 2 * (CONV_ROWS - 1) + 7;
  static constexpr int OUT_ROWS = POOL ? 8 : 16, OUT_COLS = POOL ? 15 : 32;  // outputs a tile produces
};

// tile t of the launch -> image, first output row / column
struct StemTile {
  int b, oy0, ox0;
};

template <bool POOL>
__global__ __launch_bounds__(256, 2) void stem7x7s2(float *__restrict__ out, const float *__restrict__ x,
                                                    const half8 *__restrict__ w_hi, const half8 *__restrict__ w_lo,
                                                    const float *__restrict__ w_unscale, const float *__restrict__ out_scale,
                                                    const float *__restrict__ bias, float alpha, int batch, int H, int W, int OH,
                                                    int OW, int PH, int PW, int cout, int tiles_x, int tiles_y, int n_tiles) {
  using Gm = StemGeom<POOL>;
  constexpr int PG = Gm::PG, WIN_ROWS = Gm::WIN_ROWS;
  HF_DYN_LDS;
  half8 *wl = reinterpret_cast<half8 *>(hf_dyn_lds);                                   // [2][22][64]
  _Float16 *xt = reinterpret_cast<_Float16 *>(wl + 2 * kStemGroups * 64);              // [2][3][WIN_ROWS][72]
  constexpr int PART = 3 * WIN_ROWS * kStemPitch;
  // epilogue parameters of the block's 64 channels in LDS ([2][64]: scale * 2^-k, bias): no vector-memory loads between
  // the tiles' stores, and nothing for the scheduler to hoist above the K loop (eight float4 pairs: spills)
  float *ep = reinterpret_cast<float *>(xt + 2 * PART);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int co0 = blockIdx.y * 64;
  for (int i = tid; i < kStemGroups * 64; i += 256) {
    wl[i] = w_hi[(long long)blockIdx.y * kStemGroups * 64 + i];
    wl[kStemGroups * 64 + i] = w_lo[(long long)blockIdx.y * kStemGroups * 64 + i];
  }
  if (tid < 64) {
    ep[tid] = (out_scale ? out_scale[co0 + tid] : 1.0f) * w_unscale[0];
    ep[64 + tid] = bias ? bias[co0 + tid] : 0.0f;
  }
  // window offset of the lane's K group per step: (ci * WIN_ROWS + ky) * pitch; the zero group reads group 0's data
  int koff[kStemSteps];
#pragma unroll
  for (int t = 0; t < kStemSteps; ++t) {
    const int kg = 2 * t + lh, kr = kg < 21 ? kg : 0;
    koff[t] = ((kr / 7) * WIN_ROWS + kr % 7) * kStemPitch;
  }
  const int OUT_H = POOL ? PH : OH, OUT_W = POOL ? PW : OW;
  for (int t_cur = blockIdx.x; t_cur < n_tiles; t_cur += gridDim.x) {
    StemTile T;
    int r = t_cur;
    const int tx = r % tiles_x;
    r /= tiles_x;
    T.b = r / tiles_y;
    T.oy0 = (r % tiles_y) * Gm::OUT_ROWS;
    T.ox0 = tx * Gm::OUT_COLS;
    // first conv row / column of the tile (POOL: the pooling windows start one conv pixel before 2*oy0), first input row / column
    const int cy0 = POOL ? 2 * T.oy0 - 1 : T.oy0, cx0 = POOL ? 2 * T.ox0 - 1 : T.ox0;
    const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;
    __syncthreads();  // the previous tile's window is no longer read (first tile: the weights are in)
    {
      const float *xb = x + (long long)T.b * 3 * H * W;
      bool ovf = false;
      // (requesting all of the thread's 33 elements before converting the first measured 10 % slower: the staging is
      // bound by its ~50 VALU instructions per element - index arithmetic, the saturating split - not by load latency)
      for (int i = tid; i < PART; i += 256) {
        const int c = i % kStemPitch, rr = (i / kStemPitch) % WIN_ROWS, ci = i / (kStemPitch * WIN_ROWS);
        const int iy = iy0 + rr, ix = ix0 + c;
        float v = 0.0f;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = xb[((long long)ci * H + iy) * W + ix];
        _Float16 hv, lv;
        hf_split_f16(v, hv, lv, ovf);
        xt[i] = hv;
        xt[PART + i] = lv;
      }
      hf_note_overflow(ovf);
    }
    __syncthreads();

    f32x16 acc[PG][2];
#pragma unroll
    for (int g = 0; g < PG; ++g)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[g][ct][q] = 0.0f;
    const int row_w = wave * Gm::ROW_STEP;  // the wave's first conv row inside the tile
#pragma unroll
    for (int t = 0; t < kStemSteps; ++t) {
      const int kg = 2 * t + lh;
      half8 ah[2], al[2];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        ah[ct] = wl[kg * 64 + ct * 32 + li];
        al[ct] = wl[kStemGroups * 64 + kg * 64 + ct * 32 + li];
      }
#pragma unroll
      for (int g = 0; g < PG; ++g) {
        // eight consecutive halves from column 2*li of window row 2*(row_w+g) + ky: 4-byte aligned
        const int off = koff[t] + 2 * (row_w + g) * kStemPitch + 2 * li;
        union {
          unsigned u[4];
          half8 h;
        } bh, bl;
        const unsigned *ph = reinterpret_cast<const unsigned *>(xt + off), *pl = reinterpret_cast<const unsigned *>(xt + PART + off);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          bh.u[k] = ph[k];
          bl.u[k] = pl[k];
        }
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          acc[g][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ct], bh.h, acc[g][ct], 0, 0, 0);
          acc[g][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ct], bl.h, acc[g][ct], 0, 0, 0);
          acc[g][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ct], bh.h, acc[g][ct], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // one K-step's fragments live at a time (unbounded, the scheduler hoists them all and spills)
    }

    // ---- epilogue: v = lrelu_alpha(acc * unscale * out_scale[co] + bias[co]); D layout: co = (q&3) + 8*(q>>2) + 4*lh, pixel = li
    // opaque per tile: otherwise the epilogue's per-lane offsets are computed before the K loop and live - spilled - through it
    int li_o = li, lh_o = lh, wave_o = wave;
    HF_OPAQUE_I32(li_o);
    HF_OPAQUE_I32(lh_o);
    HF_OPAQUE_I32(wave_o);
    const int row_e = wave_o * Gm::ROW_STEP;
    const int cx = cx0 + li_o;
    const bool col_ok = cx >= 0 && cx < OW;
    float *ob = out + ((long long)T.b * cout + co0) * OUT_H * OUT_W;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int c4 = ct * 32 + 8 * q4 + 4 * lh_o;
        const float4 sc4 = *reinterpret_cast<const float4 *>(ep + c4), bs4 = *reinterpret_cast<const float4 *>(ep + 64 + c4);
        const float scv[4] = {sc4.x, sc4.y, sc4.z, sc4.w};
        const float bsv[4] = {bs4.x, bs4.y, bs4.z, bs4.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float v[PG];
#pragma unroll
          for (int g = 0; g < PG; ++g) {
            const float o = fmaf(acc[g][ct][4 * q4 + k], scv[k], bsv[k]);
            v[g] = o > 0.0f ? o : o * alpha;
          }
          float *oc = ob + (long long)(c4 + k) * OUT_H * OUT_W;
          if (!POOL) {
#pragma unroll
            for (int g = 0; g < PG; ++g) {
              const int cy = cy0 + row_e + g;
              if (col_ok && cy < OH) oc[(long long)cy * OW + cx] = v[g];
            }
          } else {
            // MaxPool2d(3, 2, 1): conv pixels outside the conv plane count as -inf.  Lane li = 2j+1 is the centre of pooled
            // column ox0 + j (window = tile columns 2j .. 2j+2), rows g = 0..2 / 2..4 the windows of the wave's two pooled rows
            float hm[PG];
#pragma unroll
            for (int g = 0; g < PG; ++g) {
              const int cy = cy0 + row_e + g;
              const float m = (col_ok && cy >= 0 && cy < OH) ? v[g] : -3.402823466e38f;
              hm[g] = fmaxf(m, fmaxf(hf_lane_up(m), hf_lane_down(m)));
            }
            const int j = (li_o - 1) >> 1, px = T.ox0 + j;
            if ((li_o & 1) && li_o <= 29 && px < PW) {
              const int py = T.oy0 + 2 * wave_o;
              if (py < PH) oc[(long long)py * PW + px] = fmaxf(hm[0], fmaxf(hm[1], hm[2]));
              if (py + 1 < PH) oc[(long long)(py + 1) * PW + px] = fmaxf(hm[2], fmaxf(hm[3], hm[4]));
            }
          }
        }
      }
  }
}

}  // namespace

extern "C" unsigned long long hf_f16_overflow_count_stem(int reset) { return hf_f16_overflow_read_tu(reset); }

// LDS bytes of a launch
template <bool POOL>
static size_t stem_lds() {
  return (size_t)2 * kStemGroups * 64 * 16 + (size_t)2 * 3 * StemGeom<POOL>::WIN_ROWS * kStemPitch * 2 + 2 * 64 * sizeof(float);
}

extern "C" int hf_stem7x7s2_f16_f32(float *out, const float *x, const void *w_hi, const void *w_lo, const float *w_unscale,
                                    const float *out_scale, const float *bias, float alpha, int batch, int h, int w, int cout,
                                    int pool, void *stream) {
  if (!out || !x || !w_hi || !w_lo || !w_unscale || batch <= 0 || h <= 0 || w <= 0 || cout <= 0 || (cout % 64) ||
      !(alpha >= 0.0f && alpha <= 1.0f))
    return HF_E_INVALID;
  const int oh = (h - 1) / 2 + 1, ow = (w - 1) / 2 + 1, ph = (oh - 1) / 2 + 1, pw = (ow - 1) / 2 + 1;
  if ((long long)cout * oh * ow >= (1LL << 31)) return HF_E_INVALID;
  const int out_h = pool ? ph : oh, out_w = pool ? pw : ow;
  const int rows = pool ? StemGeom<true>::OUT_ROWS : StemGeom<false>::OUT_ROWS, cols = pool ? StemGeom<true>::OUT_COLS : StemGeom<false>::OUT_COLS;
  const int tiles_x = hf_cdiv(out_w, cols), tiles_y = hf_cdiv(out_h, rows);
  const long long n_tiles = (long long)tiles_x * tiles_y * batch;
  if (n_tiles >= (1LL << 31)) return HF_E_INVALID;
  // persistent blocks (the 45 KB of weights are copied once per block): two per CU and channel tile
  const int gx = (int)(n_tiles < 512 ? n_tiles : 512);
  const dim3 grid(gx, cout / 64);
  hipStream_t st = (hipStream_t)stream;
  const half8 *hi = static_cast<const half8 *>(w_hi), *lo = static_cast<const half8 *>(w_lo);
  if (pool) {
    hipLaunchKernelGGL(stem7x7s2<true>, grid, dim3(256), stem_lds<true>(), st, out, x, hi, lo, w_unscale, out_scale, bias, alpha, batch,
                       h, w, oh, ow, ph, pw, cout, tiles_x, tiles_y, (int)n_tiles);
  } else {
    hipLaunchKernelGGL(stem7x7s2<false>, grid, dim3(256), stem_lds<false>(), st, out, x, hi, lo, w_unscale, out_scale, bias, alpha, batch,
                       h, w, oh, ow, ph, pw, cout, tiles_x, tiles_y, (int)n_tiles);
  }
  return hf_launch_status();
}
