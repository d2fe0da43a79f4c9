// convh.hip - 3x3 stride-1 same-resolution convolution on the fp16 matrix cores
// (v_mfma_f32_32x32x16_f16: 16x the rate of the fp32 MFMA used by modconv.hip).
//
// Two operand modes, fp32 tensors in HBM and fp32 accumulation in both:
//   NTERMS = 3  "f16x3": every fp32 operand v is split into hi = fp16(v) and
//               lo = fp16(v - hi) (22 significant bits together) and the product a*b is
//               formed as hi*hi + hi*lo + lo*hi - three fp16 MFMAs whose products are exact
//               in the fp32 accumulator.  The dropped lo*lo term and the 22-bit operands
//               leave a relative error of ~5e-7 per product, the same class as an fp32
//               reassociation; the cost is 3/16 of the fp32 MFMA time.
//               Range: |s*x| and |w| must stay below 65504 (fp16); StyleGAN2 activations
//               are O(1..1e3).
//   NTERMS = 1  "f16": operands rounded to fp16 (BASELINE.json configs[4]: fp16 MFMA with
//               fp32 demodulation / accumulation), relative error ~5e-4 per product.
//
// Same contraction as modconv.hip: y = d * sum wt[tap,ci,co] * (s[b,ci] x[b,ci,p+tap]),
// epilogue shared (store_tile).  Layouts are K-contiguous because an MFMA operand is
// 8 consecutive input channels per lane:
//   weights (global, prepared once):  [chunk16][tap][kgroup 2][cout][8 halves], hi and lo;
//     a stage is 9*2 rows of CT*16 B -> 1 KiB global_load_lds pieces, A fragment =
//     ds_read_b128 of 32 consecutive cout (conflict free);
//   activations (LDS): [kgroup 2][halo pixel][8 halves], hi and lo; B fragment = ds_read_b128
//     of 32 consecutive pixels of a tile row.  A thread stages one (pixel, kgroup) item:
//     8 coalesced plane loads (lanes = consecutive pixels), * s, split, one 16 B LDS write
//     per half.
// Pipeline: during chunk c the weights of chunk c+1 arrive by DMA and its activations travel
// through registers (loaded in the first tap-step, converted in the last ones); one barrier per
// chunk.  LDS-DMA and register loads retire through different paths, so vmcnt cannot order one
// against the other: everything is issued early in the chunk and drained at its barrier.
#include "conv_common.h"

using namespace hf_detail;

#ifndef HF_H_VARIANT
#define HF_H_VARIANT 0
#endif
#if HF_H_VARIANT & 1
#define HF_H_GLDS hf_glds16
#define HF_H_BARRIER() __syncthreads()
#else
#define HF_H_GLDS hf_glds16_raw
#define HF_H_BARRIER() hf_barrier_keep_young<0>()
#endif

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
constexpr int KH = 16;  // input channels per stage = K of one MFMA

template <int NTERMS, int CT_TILES, int PG, int WAVES_CO, int WAVES_PX, bool MOD>
__global__ __launch_bounds__(64 * WAVES_CO * WAVES_PX) void conv_mfma_h(const ConvParams P,
                                                                          const _Float16 *__restrict__ wth,
                                                                          const _Float16 *__restrict__ wtl) {
  constexpr int NW = WAVES_CO * WAVES_PX;
  constexpr int NT = 64 * NW;
  constexpr int CT = 32 * CT_TILES * WAVES_CO;
  constexpr int TH = PG * WAVES_PX;  // tile = TH rows x 32 columns
  constexpr int HP = TH + 2, WP = 34, NPIX = HP * WP;
  constexpr int NPART = (NTERMS == 3) ? 2 : 1;          // hi (+ lo)
  constexpr int W_UNITS = 9 * 2 * CT;                   // 16-byte units of one weight part per stage
  constexpr int X_UNITS = 2 * NPIX;                     // 16-byte units of one activation part per stage
  constexpr int BUF_UNITS = NPART * (W_UNITS + X_UNITS);
  constexpr int N_WPIECE = NPART * W_UNITS / 64;        // 1 KiB DMA pieces per stage
  constexpr int ND = (N_WPIECE + NW - 1) / NW;          // per wave
  constexpr int DMA_PER_STEP = (ND + 2) / 3;            // all weight DMAs in the first three tap-steps
  constexpr int XE = (X_UNITS + NT - 1) / NT;           // (pixel, kgroup) items per thread per stage
  static_assert(W_UNITS % 64 == 0, "weight part must be whole 1 KiB pieces");

  HF_DYN_LDS;
  half8 *lds = reinterpret_cast<half8 *>(hf_dyn_lds);               // [2][BUF_UNITS] 16-byte units
  float *sl = reinterpret_cast<float *>(lds + 2 * BUF_UNITS);       // s[cin]
  // buffer layout (units): [W hi][W lo][X hi][X lo]
  constexpr int OFF_WL = W_UNITS, OFF_XH = NPART * W_UNITS, OFF_XL = NPART * W_UNITS + X_UNITS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int li = lane & 31;
  const int lh = lane >> 5;  // k group of the lane: input channels 8*lh .. 8*lh+7 of the stage
  const int wave_co = (wave / WAVES_PX) * (32 * CT_TILES);
  const int wave_pg = (wave % WAVES_PX) * PG;
  const int co0 = blockIdx.y * CT;

  const TileGeom G = P.g[0];
  int t = blockIdx.x;
  const int tx = t % G.tiles_x;
  t /= G.tiles_x;
  const int ty = t % G.tiles_y;
  const int b0 = t / G.tiles_y;
  const int ty0 = ty * TH, tx0 = tx * 32;
  const long long plane = (long long)P.h * P.w;
  const float *xb = P.x + (long long)b0 * P.cin * plane;

  if (MOD)
    for (int i = tid; i < P.cin; i += NT) sl[i] = P.s[(long long)b0 * P.s_bstride + i];

  // staging items of this thread: (pixel, kgroup) -> plane offset (-1: zero) and LDS unit
  int e_src[XE], e_kg[XE];
#pragma unroll
  for (int e = 0; e < XE; ++e) {
    const int i = tid + e * NT;
    e_src[e] = -2;  // -2: no item, -1: item outside the image (zero fill)
    e_kg[e] = 0;
    if (i < X_UNITS) {
      const int kg = i / NPIX, pix = i - kg * NPIX;
      const int hy = pix / WP, hx = pix - hy * WP;
      const int ys = ty0 + hy - 1, xc = tx0 + hx - 1;
      e_kg[e] = kg;
      e_src[e] = (ys >= 0 && ys < P.h && xc >= 0 && xc < P.w) ? (int)((long long)ys * P.w + xc) : -1;
    }
  }

  auto dma_piece = [&](int i, int chunk, half8 *buf) {
    const int pc = wave + i * NW;
    if (pc < N_WPIECE) {
      const int part = pc / (W_UNITS / 64), q = pc % (W_UNITS / 64);
      const int u = q * 64 + lane;            // unit inside the part: (tap*2 + kg)*CT + co
      const int row = u / CT, col = u % CT;   // row = tap*2 + kg
      const _Float16 *src = (part ? wtl : wth) + (((long long)chunk * 18 + row) * P.cout + co0 + col) * 8;
      HF_H_GLDS(reinterpret_cast<const float *>(src), reinterpret_cast<float *>(buf + part * W_UNITS + q * 64));
    }
  };

  float xr[1][XE][8];  // raw activations of the next stage, in flight for one whole chunk
  // uniform chunk base + per-lane offset; halo items outside the image load element 0 of the
  // chunk instead (unconditional loads: no branches in the pipeline) and are zeroed at conversion
  const int iplane = (int)plane;
  auto load_item = [&](int set, int e, int chunk) {
    const float *xc = xb + (long long)chunk * KH * plane;
    const int off = (e_src[e] >= 0) ? e_kg[e] * 8 * iplane + e_src[e] : 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) xr[set][e][k] = xc[off + k * iplane];
  };
  auto convert_item = [&](int set, int e, int chunk, half8 *buf) {
    if (e_src[e] == -2) return;
    const int i = tid + e * NT;
    half8 hi, lo;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float v = (e_src[e] >= 0) ? xr[set][e][k] : 0.0f;
      if (MOD) v *= sl[chunk * KH + e_kg[e] * 8 + k];
      // hi and lo must both derive from the fp32-ROUNDED product: left alone, hipcc stores
      // hi = fp16(fp32(x*s)) but subtracts v_fma_mixlo_f16's fp16(x*s unrounded); at an fp16
      // rounding tie of the fp32 product the two differ by one fp16 ulp (seen on hardware).
      HF_OPAQUE_F32(v);
      const _Float16 hv = (_Float16)v;
      hi[k] = hv;
      lo[k] = (_Float16)(v - (float)hv);
    }
    buf[OFF_XH + i] = hi;
    if (NTERMS == 3) buf[OFF_XL + i] = lo;
  };

  int pixoff[PG];
#pragma unroll
  for (int g = 0; g < PG; ++g) pixoff[g] = (wave_pg + g) * WP + li;  // one 32-pixel row per group

  f32x16 acc[1][CT_TILES][PG];
#pragma unroll
  for (int ct = 0; ct < CT_TILES; ++ct)
#pragma unroll
    for (int g = 0; g < PG; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][ct][g][r] = 0.0f;

  const int nchunks = P.cin / KH;
  __syncthreads();  // sl visible
  // prologue: stage 0 resident, activations of stage 1 in flight
#pragma unroll
  for (int i = 0; i < ND; ++i) dma_piece(i, 0, lds);
#pragma unroll
  for (int e = 0; e < XE; ++e) load_item(0, e, 0);
#pragma unroll
  for (int e = 0; e < XE; ++e) convert_item(0, e, 0, lds);
  HF_H_BARRIER();

  // one chunk = 9 tap-steps; side work spread over the steps: step 0 issues the activation
  // loads of chunk c+1 into registers, steps 0-2 its weight DMAs, the last steps convert the
  // activations (6+ steps of MFMA time after their loads) into the other buffer
  for (int c = 0; c < nchunks; ++c) {
    const int cur = c & 1;
    half8 *buf = lds + cur * BUF_UNITS, *nbuf = lds + (cur ^ 1) * BUF_UNITS;
    const bool more1 = c + 1 < nchunks;
    const half8 *a_hi = buf + lh * CT + wave_co + li;  // + tap*2*CT + ct*32
    const half8 *b_hi = buf + OFF_XH + lh * NPIX;      // + pixoff + toff
    // fragments of tap+1 are fetched from LDS while the MFMAs of tap run
    half8 ah[2][CT_TILES], al[2][CT_TILES], bh[2][PG], bl[2][PG];
    auto fetch = [&](int slot, int tap) {
      const int toff = (tap / 3) * WP + tap % 3;
#pragma unroll
      for (int ct = 0; ct < CT_TILES; ++ct) {
        ah[slot][ct] = a_hi[tap * 2 * CT + ct * 32];
        if (NTERMS == 3) al[slot][ct] = a_hi[OFF_WL + tap * 2 * CT + ct * 32];
      }
#pragma unroll
      for (int g = 0; g < PG; ++g) {
        bh[slot][g] = b_hi[pixoff[g] + toff];
        if (NTERMS == 3) bl[slot][g] = b_hi[X_UNITS + pixoff[g] + toff];
      }
    };
    fetch(0, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int sl_ = tap & 1;
#if HF_H_VARIANT & 2
      fetch(sl_, tap);
#else
      if (tap + 1 < 9) fetch(sl_ ^ 1, tap + 1);
#endif
      // ---- side work of this step ----
      if (more1) {
#pragma unroll
        for (int i = 0; i < ND; ++i)
          if (i / DMA_PER_STEP == tap) dma_piece(i, c + 1, nbuf);
      }
      if (more1 && tap == 0) {
#pragma unroll
        for (int e = 0; e < XE; ++e) load_item(0, e, c + 1);
      }
      if (more1 && tap >= 9 - XE) convert_item(0, tap - (9 - XE), c + 1, nbuf);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ct = 0; ct < CT_TILES; ++ct)
#pragma unroll
        for (int g = 0; g < PG; ++g)
          acc[0][ct][g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[sl_][ct], bh[sl_][g], acc[0][ct][g], 0, 0, 0);
      if (NTERMS == 3) {
#pragma unroll
        for (int ct = 0; ct < CT_TILES; ++ct)
#pragma unroll
          for (int g = 0; g < PG; ++g)
            acc[0][ct][g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[sl_][ct], bl[sl_][g], acc[0][ct][g], 0, 0, 0);
#pragma unroll
        for (int ct = 0; ct < CT_TILES; ++ct)
#pragma unroll
          for (int g = 0; g < PG; ++g)
            acc[0][ct][g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[sl_][ct], bh[sl_][g], acc[0][ct][g], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // next stage complete (DMA landed, conversions written), current one free
    HF_H_BARRIER();
  }

  store_tile<CT_TILES, PG, false>(P, G, GroupOfs{0, 0, 0, 0, 0}, acc, co0 + wave_co, wave_pg, li, lh, ty0, tx0, b0);
}

// fp32 prepared weights wt[tap][ci][co] -> hi / lo halves in [chunk16][tap][kg][co][8]
__global__ __launch_bounds__(256) void split_weights(_Float16 *__restrict__ wth, _Float16 *__restrict__ wtl,
                                                     const float *__restrict__ wt, int cin, int cout) {
  const long long n = 9LL * cin * cout;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int co = (int)(i % cout);
    const long long r = i / cout;
    const int ci = (int)(r % cin), tap = (int)(r / cin);
    const float v = wt[i];
    const _Float16 h = (_Float16)v;  // plain load: nothing to contract with
    const long long dst = ((((long long)(ci / 16) * 9 + tap) * 2 + (ci % 16) / 8) * cout + co) * 8 + (ci % 8);
    wth[dst] = h;
    wtl[dst] = (_Float16)(v - (float)h);
  }
}

template <int NTERMS, int CT_TILES, int PG, int WAVES_CO, int WAVES_PX>
int launch_h(ConvParams &P, const _Float16 *wth, const _Float16 *wtl, hipStream_t st) {
  constexpr int NT = 64 * WAVES_CO * WAVES_PX;
  constexpr int CT = 32 * CT_TILES * WAVES_CO;
  constexpr int TH = PG * WAVES_PX, HP = TH + 2;
  constexpr int NPART = (NTERMS == 3) ? 2 : 1;
  if (P.cin % KH || P.cout % CT || P.stride != 1 || P.t || P.groups > 1) return HF_E_INVALID;
  if (P.w < 32 || P.h < TH) return HF_E_INVALID;
  if ((long long)P.cin * P.h * P.w >= (1LL << 31)) return HF_E_INVALID;
  P.splits = 1;
  P.n_geom = 1;
  TileGeom g{};
  g.y0 = 0; g.x0 = 0; g.dh = P.h; g.dw = P.w;
  g.lg_tw = 5; g.lg_th = ilog2(TH); g.lg_nb = 0;
  g.tiles_x = hf_cdiv(P.w, 32); g.tiles_y = hf_cdiv(P.h, TH); g.tiles_b = P.batch; g.first_block = 0;
  P.g[0] = g;
  const size_t lds = (size_t)2 * NPART * (9 * 2 * CT + 2 * HP * 34) * 16 + (P.s ? P.cin * sizeof(float) : 0);
  if (lds > 160 * 1024) return HF_E_INVALID;
  dim3 grid(geom_blocks(g), P.cout / CT);
  if (grid.y > 65535) return HF_E_INVALID;
  if (P.s)
    hipLaunchKernelGGL((conv_mfma_h<NTERMS, CT_TILES, PG, WAVES_CO, WAVES_PX, true>), grid, dim3(NT), lds, st, P, wth, wtl);
  else
    hipLaunchKernelGGL((conv_mfma_h<NTERMS, CT_TILES, PG, WAVES_CO, WAVES_PX, false>), grid, dim3(NT), lds, st, P, wth, wtl);
  return hf_launch_status();
}

}  // namespace

namespace hf_detail {

int g_force_h = 0;

int launch_conv_h(ConvParams &P, int nterms, const void *wth, const void *wtl, hipStream_t st) {
  const _Float16 *h = static_cast<const _Float16 *>(wth), *l = static_cast<const _Float16 *>(wtl);
  if (!h || (nterms == 3 && !l)) return HF_E_INVALID;
  // 51: 64 co x 256 px (8 rows), 2 co-waves x 4 pixel-waves, 1x2 MFMA tiles per wave
  // 52: 64 co x 512 px (16 rows), 8 pixel-waves, 2x2 MFMA tiles per wave: 0.67 LDS fragment
  //     reads per MFMA instead of 2 (fewer issue slots beside the MFMAs), needs >= 256 such blocks
  // 53: 32 co x 512 px (16 rows), 8 pixel-waves, 1x2 tiles: layers with cout % 64 != 0 (1024^2: 32)
  int cfg = g_force_h;
  if (cfg == 0) {
    const long long blocks52 = (long long)P.batch * hf_cdiv(P.h, 16) * hf_cdiv(P.w, 32) * (P.cout / 64);
    if (P.cout % 64) cfg = 53;
    else cfg = (P.h >= 16 && blocks52 >= 256) ? 52 : 51;
  }
  if ((cfg == 52 || cfg == 53) && P.h < 16) cfg = 51;
  int rc;
  if (cfg == 53) rc = (nterms == 3) ? launch_h<3, 1, 2, 1, 8>(P, h, l, st) : launch_h<1, 1, 2, 1, 8>(P, h, l, st);
  else if (cfg == 52) rc = (nterms == 3) ? launch_h<3, 2, 2, 1, 8>(P, h, l, st) : launch_h<1, 2, 2, 1, 8>(P, h, l, st);
  else rc = (nterms == 3) ? launch_h<3, 1, 2, 2, 4>(P, h, l, st) : launch_h<1, 1, 2, 2, 4>(P, h, l, st);
  if (rc == HF_OK) note_path(5, cfg);
  return rc;
}

}  // namespace hf_detail

extern "C" int hf_conv_split_weights_f16(void *wt_hi, void *wt_lo, const float *wt, int cin, int cout, void *stream) {
  if (!wt_hi || !wt_lo || !wt || cin <= 0 || cout <= 0 || (cin % 16)) return HF_E_INVALID;
  long long n = 9LL * cin * cout;
  long long g = (n + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(split_weights, dim3((int)g), dim3(256), 0, (hipStream_t)stream, static_cast<_Float16 *>(wt_hi),
                     static_cast<_Float16 *>(wt_lo), wt, cin, cout);
  return hf_launch_status();
}

extern "C" int hf_modconv3x3_f16_f32(float *out, const float *x, const void *wt_hi, const void *wt_lo, int nterms,
                                     const float *s, const float *d, const float *noise, const float *noise_w,
                                     long long noise_bstride, const float *bias, int batch, int cin, int cout, int h,
                                     int w, float alpha, float scale, void *stream) {
  if (!out || !x || !wt_hi || batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0 || (noise && !noise_w) ||
      (nterms != 1 && nterms != 3))
    return HF_E_INVALID;
  ConvParams P{};
  P.out = out; P.x = x; P.s = s; P.d = d; P.noise = noise; P.noise_w = noise_w; P.bias = bias;
  P.s_bstride = cin; P.d_bstride = cout;
  P.groups = 1;
  P.noise_bstride = noise_bstride;
  P.batch = batch; P.cin = cin; P.cout = cout; P.h = h; P.w = w; P.out_h = h; P.out_w = w; P.out_wv = w;
  P.stride = 1;
  P.act = bias ? ACT_LRELU : ACT_NONE;
  P.alpha = alpha; P.scale = scale;
  return launch_conv_h(P, nterms, wt_hi, wt_lo, (hipStream_t)stream);
}
