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
#define HF_H_BARRIER() __syncthreads()
#else
#define HF_H_BARRIER() hf_barrier_keep_young<0>()
#endif

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
constexpr int KH = 16;  // input channels per stage = K of one MFMA

// Halo pixels the LDS activation tile is sized for: PT-pixel tiles of 32-pixel rows, and (UP)
// the 2-row / 2-column rim tiles of make_geom(one_image).
template <int PT, bool UP>
constexpr int halo_pixels_max() {
  constexpr int main_tile = (PT / 32 + (UP ? 1 : 2)) * (32 + (UP ? 1 : 2));
  constexpr int rim_tile = 3 * (PT / 2 + 1);
  return (UP && rim_tile > main_tile) ? rim_tile : main_tile;
}

// UP: the transposed (stride 2) conv of the upsampling StyledConv, as in modconv.hip: 4 output
// phases (pr,pc) = (ky&1, kx&1) per input position (Y,X) of the (h+1)x(w+1) phase domain, each
// tap feeding exactly one phase from x[Y - (ky==2), X - (kx==2)]; no zero-insertion flops.
template <int NTERMS, int CT_TILES, int PG, int WAVES_CO, int WAVES_PX, bool MOD, bool UP>
__global__ __launch_bounds__(64 * WAVES_CO * WAVES_PX) void conv_mfma_h(const ConvParams P,
                                                                          const _Float16 *__restrict__ wth,
                                                                          const _Float16 *__restrict__ wtl) {
  constexpr int NW = WAVES_CO * WAVES_PX;
  constexpr int NT = 64 * NW;
  constexpr int CT = 32 * CT_TILES * WAVES_CO;
  constexpr int PT = 32 * PG * WAVES_PX;  // pixels (UP: phase-domain positions) per tile
  constexpr int NPH = UP ? 4 : 1;
  constexpr int HALO = UP ? 1 : 2;
  constexpr int NPIX = halo_pixels_max<PT, UP>();
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

  int gi = 0;  // tile family (uniform): interior, or the rim row / column of the transposed conv
  if (P.n_geom > 1 && (int)blockIdx.x >= P.g[1].first_block) gi = 1;
  if (P.n_geom > 2 && (int)blockIdx.x >= P.g[2].first_block) gi = 2;
  const TileGeom G = P.g[gi];
  int t = blockIdx.x - G.first_block;
  const int tx = t % G.tiles_x;
  t /= G.tiles_x;
  const int ty = t % G.tiles_y;
  const int b0 = t / G.tiles_y;  // one image per tile
  const int tw = 1 << G.lg_tw, th = 1 << G.lg_th;
  const int wp = tw + HALO, xs = (th + HALO) * wp;  // halo tile, first pixel (ty0-1, tx0-1)
  const int ty0 = G.y0 + ty * th, tx0 = G.x0 + tx * tw;
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
    const int kg = i / NPIX, pix = i - kg * NPIX;
    if (i < X_UNITS && pix < xs) {
      const int hy = pix / wp, hx = pix - hy * wp;
      const int ys = ty0 + hy - 1, xc = tx0 + hx - 1;
      e_kg[e] = kg;
      e_src[e] = (ys >= 0 && ys < P.h && xc >= 0 && xc < P.w) ? (int)((long long)ys * P.w + xc) : -1;
    }
  }

  // weight stage of `chunk`: uniform base + per-lane byte offset ((tap*2+kg)*cout + co0 + col)*16
  auto dma_piece = [&](int i, int chunk, half8 *buf) {
    const int pc = wave + i * NW;
    if (pc < N_WPIECE) {
      const int part = pc / (W_UNITS / 64), q = pc % (W_UNITS / 64);
      const int u = q * 64 + lane;            // unit inside the part: (tap*2 + kg)*CT + co
      const int row = u / CT, col = u % CT;   // row = tap*2 + kg
      int off = (row * P.cout + co0 + col) * 16;
      HF_OPAQUE_I32(off);
      const _Float16 *src = (part ? wtl : wth) + (long long)chunk * 18 * P.cout * 8;
#if HF_H_VARIANT & 1
      hf_glds16(reinterpret_cast<const float *>(reinterpret_cast<const char *>(src) + off),
                reinterpret_cast<float *>(buf + part * W_UNITS + q * 64));
#else
      hf_glds16_raw_s(src, (unsigned)off, reinterpret_cast<float *>(buf + part * W_UNITS + q * 64));
#endif
    }
  };

  float xr[1][XE][8];  // raw activations of the next stage, in flight for one whole chunk
  // uniform chunk base + per-lane offset; halo items outside the image load element 0 of the
  // chunk instead (unconditional loads: no branches in the pipeline) and are zeroed at conversion
  const int iplane = (int)plane;
  auto load_item = [&](int set, int e, int chunk) {
    const char *xc = reinterpret_cast<const char *>(xb + (long long)chunk * KH * plane);  // uniform
    int off = (e_src[e] >= 0) ? e_kg[e] * 8 * iplane + e_src[e] : 0;
    HF_OPAQUE_I32(off);  // addresses recomputed per chunk, not kept live as 8 hoisted 64-bit pairs
#pragma unroll
    for (int k = 0; k < 8; ++k)
      xr[set][e][k] = *reinterpret_cast<const float *>(xc + (unsigned)((off + k * iplane) * 4));
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

  // halo-tile unit of the lane's pixel of group g, per tap row (ky*wp; UP: rows 1 and 0)
  constexpr int NROW = UP ? 2 : 3;
  int pixrow[PG][NROW];
#pragma unroll
  for (int g = 0; g < PG; ++g) {
    const int p = (wave_pg + g) * 32 + li;
    const int po = ((p >> G.lg_tw) & (th - 1)) * wp + (p & (tw - 1));
#pragma unroll
    for (int r = 0; r < NROW; ++r) pixrow[g][r] = po + r * wp;
  }

  f32x16 acc[NPH][CT_TILES][PG];
#pragma unroll
  for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
    for (int ct = 0; ct < CT_TILES; ++ct)
#pragma unroll
      for (int g = 0; g < PG; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ph][ct][g][r] = 0.0f;

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
    constexpr int NSLOT = UP ? 1 : 2;  // UP: 128 accumulator registers leave no room for a second set
    half8 ah[NSLOT][CT_TILES], al[NSLOT][CT_TILES], bh[NSLOT][PG], bl[NSLOT][PG];
    auto fetch = [&](int slot, int tap) {
      const int ky = tap / 3, kx = tap % 3;
      const int brow = UP ? (ky == 2 ? 0 : 1) : ky, bcol = UP ? (kx == 2 ? 0 : 1) : kx;
#pragma unroll
      for (int ct = 0; ct < CT_TILES; ++ct) {
        ah[slot][ct] = a_hi[tap * 2 * CT + ct * 32];
        if (NTERMS == 3) al[slot][ct] = a_hi[OFF_WL + tap * 2 * CT + ct * 32];
      }
#pragma unroll
      for (int g = 0; g < PG; ++g) {
        bh[slot][g] = b_hi[pixrow[g][brow] + bcol];
        if (NTERMS == 3) bl[slot][g] = b_hi[X_UNITS + pixrow[g][brow] + bcol];
      }
    };
    fetch(0, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int sl_ = UP ? 0 : (tap & 1);
      if (UP) {
        if (tap > 0) fetch(0, tap);
      } else if (tap + 1 < 9) {
        fetch(sl_ ^ 1, tap + 1);
      }
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
      const int ph = UP ? (((tap / 3) & 1) * 2 + ((tap % 3) & 1)) : 0;
#pragma unroll
      for (int ct = 0; ct < CT_TILES; ++ct)
#pragma unroll
        for (int g = 0; g < PG; ++g)
          acc[ph][ct][g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[sl_][ct], bh[sl_][g], acc[ph][ct][g], 0, 0, 0);
      if (NTERMS == 3) {
#pragma unroll
        for (int ct = 0; ct < CT_TILES; ++ct)
#pragma unroll
          for (int g = 0; g < PG; ++g)
            acc[ph][ct][g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[sl_][ct], bl[sl_][g], acc[ph][ct][g], 0, 0, 0);
#pragma unroll
        for (int ct = 0; ct < CT_TILES; ++ct)
#pragma unroll
          for (int g = 0; g < PG; ++g)
            acc[ph][ct][g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[sl_][ct], bh[sl_][g], acc[ph][ct][g], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // next stage complete (DMA landed, conversions written), current one free
    HF_H_BARRIER();
  }

  store_tile<CT_TILES, PG, UP>(P, G, GroupOfs{0, 0, 0, 0, 0}, acc, co0 + wave_co, wave_pg, li, lh, ty0, tx0, b0);
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

template <int NTERMS, int CT_TILES, int PG, int WAVES_CO, int WAVES_PX, bool UP>
int launch_h(ConvParams &P, const _Float16 *wth, const _Float16 *wtl, hipStream_t st) {
  constexpr int NT = 64 * WAVES_CO * WAVES_PX;
  constexpr int CT = 32 * CT_TILES * WAVES_CO;
  constexpr int PT = 32 * PG * WAVES_PX;
  constexpr int NPIX = halo_pixels_max<PT, UP>();
  constexpr int NPART = (NTERMS == 3) ? 2 : 1;
  if (P.cin % KH || P.cout % CT || P.stride != 1 || P.t || P.groups > 1) return HF_E_INVALID;
  if ((long long)P.cin * P.h * P.w >= (1LL << 31)) return HF_E_INVALID;
  P.splits = 1;
  P.n_geom = 1;
  P.g[0] = make_geom(0, 0, P.h, P.w, P.batch, PT, 0);
  int nblocks = geom_blocks(P.g[0]);
  if (UP) {  // + the Y = h row (incl. corner) and the X = w column of the (h+1)x(w+1) phase domain
    P.n_geom = 3;
    P.g[1] = make_geom(P.h, 0, 1, P.w + 1, P.batch, PT, nblocks, true);
    nblocks += geom_blocks(P.g[1]);
    P.g[2] = make_geom(0, P.w, P.h, 1, P.batch, PT, nblocks, true);
    nblocks += geom_blocks(P.g[2]);
  }
  for (int i = 0; i < P.n_geom; ++i) {
    // one image per tile, whole PT-pixel tiles, halo within the LDS tile
    if (P.g[i].lg_nb != 0 || (1 << (P.g[i].lg_tw + P.g[i].lg_th)) != PT) return HF_E_INVALID;
    if (geom_xs(P.g[i], 1, UP ? 1 : 2) > NPIX) return HF_E_INVALID;
  }
  const size_t lds = (size_t)2 * NPART * (9 * 2 * CT + 2 * NPIX) * 16 + (P.s ? P.cin * sizeof(float) : 0);
  if (lds > 160 * 1024) return HF_E_INVALID;
  dim3 grid(nblocks, P.cout / CT);
  if (grid.y > 65535) return HF_E_INVALID;
  if (P.s)
    hipLaunchKernelGGL((conv_mfma_h<NTERMS, CT_TILES, PG, WAVES_CO, WAVES_PX, true, UP>), grid, dim3(NT), lds, st, P, wth,
                       wtl);
  else
    hipLaunchKernelGGL((conv_mfma_h<NTERMS, CT_TILES, PG, WAVES_CO, WAVES_PX, false, UP>), grid, dim3(NT), lds, st, P, wth,
                       wtl);
  return hf_launch_status();
}

}  // namespace

namespace hf_detail {

int g_force_h = 0;

int launch_conv_h(ConvParams &P, int nterms, bool up, const void *wth, const void *wtl, hipStream_t st) {
  const _Float16 *h = static_cast<const _Float16 *>(wth), *l = static_cast<const _Float16 *>(wtl);
  if (!h || (nterms == 3 && !l)) return HF_E_INVALID;
  int cfg, rc;
  if (up) {
    // 61: 64 co x 256 positions x 4 phases, 2 co-waves x 4 pixel-waves, 1x2 MFMA tiles per phase
    // 63: 32 co x 512 positions x 4 phases, 8 pixel-waves (cout % 64 != 0: the 1024^2 layer)
    cfg = (P.cout % 64) ? 63 : 61;
    if (cfg == 63) rc = (nterms == 3) ? launch_h<3, 1, 2, 1, 8, true>(P, h, l, st) : launch_h<1, 1, 2, 1, 8, true>(P, h, l, st);
    else rc = (nterms == 3) ? launch_h<3, 1, 2, 2, 4, true>(P, h, l, st) : launch_h<1, 1, 2, 2, 4, true>(P, h, l, st);
    if (rc == HF_OK) note_path(5, cfg);
    return rc;
  }
  // 51: 64 co x 256 px (8 rows), 2 co-waves x 4 pixel-waves, 1x2 MFMA tiles per wave
  // 52: 64 co x 512 px (16 rows), 8 pixel-waves, 2x2 MFMA tiles per wave: 0.67 LDS fragment
  //     reads per MFMA instead of 2 (fewer issue slots beside the MFMAs), needs >= 256 such blocks
  // 53: 32 co x 512 px (16 rows), 8 pixel-waves, 1x2 tiles: layers with cout % 64 != 0 (1024^2: 32)
  cfg = g_force_h;
  if (cfg == 0) {
    const long long blocks52 = (long long)P.batch * hf_cdiv(P.h, 16) * hf_cdiv(P.w, 32) * (P.cout / 64);
    if (P.cout % 64) cfg = 53;
    else cfg = (P.h * P.w >= 512 && blocks52 >= 256) ? 52 : 51;
  }
  if (cfg == 53) rc = (nterms == 3) ? launch_h<3, 1, 2, 1, 8, false>(P, h, l, st) : launch_h<1, 1, 2, 1, 8, false>(P, h, l, st);
  else if (cfg == 52) rc = (nterms == 3) ? launch_h<3, 2, 2, 1, 8, false>(P, h, l, st) : launch_h<1, 2, 2, 1, 8, false>(P, h, l, st);
  else rc = (nterms == 3) ? launch_h<3, 1, 2, 2, 4, false>(P, h, l, st) : launch_h<1, 1, 2, 2, 4, false>(P, h, l, st);
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
  return launch_conv_h(P, nterms, false, wt_hi, wt_lo, (hipStream_t)stream);
}

extern "C" int hf_modconv3x3_up_f16_f32(float *tmp, const float *x, const void *wt_hi, const void *wt_lo, int nterms,
                                        const float *s, const float *d, int batch, int cin, int cout, int h, int w,
                                        int tmp_pitch, void *stream) {
  if (!tmp || !x || !wt_hi || batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0 || tmp_pitch < 2 * w + 1 ||
      (nterms != 1 && nterms != 3))
    return HF_E_INVALID;
  ConvParams P{};
  P.out = tmp; P.x = x; P.s = s; P.d = d;
  P.s_bstride = cin; P.d_bstride = cout;
  P.groups = 1;
  P.batch = batch; P.cin = cin; P.cout = cout; P.h = h; P.w = w; P.out_h = 2 * h + 1; P.out_w = tmp_pitch;
  P.out_wv = 2 * w + 1;
  P.stride = 1;
  return launch_conv_h(P, nterms, true, wt_hi, wt_lo, (hipStream_t)stream);
}
