// convh.hip - 3x3 stride-1 same-resolution convolution on the fp16 matrix cores
// (v_mfma_f32_32x32x16_f16: 16x the rate of the fp32 MFMA used by modconv.hip).
//
// Two operand modes, fp32 tensors in HBM and fp32 accumulation in both:
//   NTERMS = 3  "f16x3": every fp32 operand v is split into hi = fp16(v) and
//               lo = fp16(v - hi) (22 significant bits together) and the product a*b is
//               formed as hi*hi + hi*lo + lo*hi - three fp16 MFMAs whose products are exact
//               in the fp32 accumulator.  The dropped lo*lo term and the 22-bit operands
//               leave a relative error of 2^-22 per operand (absolute 2^-25 below 2^-3, see
//               hf_common.h), the same class as an fp32 reassociation; the cost is 3/16 of the
//               fp32 MFMA time.  Range: weights are pre-scaled by a power of two and styles
//               normalised to [1, 2) by the producers (split_weights, style.hip), both parts
//               saturate at +-65504 (values up to 131008 are carried, larger ones clamp and are
//               counted: hf_f16_overflow_count).
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
#define HF_WANT_F16_SPLIT
#include <type_traits>

#include "conv_common.h"

using namespace hf_detail;

#define HF_H_BARRIER() hf_barrier_keep_young<0>()
#ifndef HF_H_SPLIT_STORE16
#define HF_H_SPLIT_STORE16 0  // fused upsampling epilogue: 1 = split output as ONE 16-byte store per lane (v_permlane32_swap of the half-waves) instead of two 8-byte stores; measured equal (587 vs 595 us on the 1024^2 layer): the epilogue is not store-bound
#endif
#ifndef HF_H_SPLIT_STORE_PAIR
#define HF_H_SPLIT_STORE_PAIR 1  // fused upsampling epilogue: the half-waves trade their halves of the two pixels (2X, 2X+1) of a position (v_permlane32_swap), every lane then owns ONE whole 16-byte unit: a store instruction writes 1 KiB contiguous instead of every other 8 bytes of it (half the write requests, half the store instructions); 0 = two 8-byte stores per pixel (A/B builds)
#endif
#ifndef HF_H_SWAP_XY
#define HF_H_SWAP_XY 1  // 0: always the (tile walkers, cout tiles) grid (A/B builds)
#endif
#ifndef HF_H_PINGPONG
#define HF_H_PINGPONG 1  // 0: the one-phase K loop (side work of a tap-step, then its MFMAs, all eight waves in lock-step) for A/B builds
#endif
#ifndef HF_H_EPI_ABLATE
#define HF_H_EPI_ABLATE 0  // timing experiments on the fused upsampling epilogue only (WRONG results): 1 no output stores, 2 no lo part (one conversion per element), 4 no cross-wave exchange (no LDS round trip, no barriers), 8 no vertical taps
#endif
#ifndef HF_H_PP_ROLES
#define HF_H_PP_ROLES 0  // ping-pong K loop (measured neutral: fused layers 552-569 vs 592 us, 512->512 @64^2 425 vs 410, r06j - off): 1 = the half that idles FIRST in a stage (waves 4-7, phase A) issues ALL activation copies of the next stage (HBM / Infinity-Cache latency: they get the whole stage to land), the other half (phase B) only the weight copies (L2 hits); 0 = every wave issues its share of both in its idle phase (A/B builds)
#endif
#ifndef HF_H_LATE_TABLES
#define HF_H_LATE_TABLES 0  // 1: the next image's epilogue tables loaded at the head of the tile into registers and written to LDS after its K loop, no barrier (measured neutral: 1706.7 vs 1707.0 img/s, r06al - off)
#endif
#ifndef HF_H_ILV
#define HF_H_ILV 1  // ping-pong K loop: 1 = the LDS fragment reads of the NEXT tap are issued between the MFMAs of the current one (sched_group_barrier: one read behind each MFMA) instead of in front of them - in its turn on the pipe a wave is alone on its SIMD, nothing else covers the ~150 cycles the eight ds_read_b128 take to issue (profiles/r06af_trace_same_res_64ch.txt: step -> mfma 200 ticks per tap beside 500 of MFMAs)
#endif
#ifndef HF_H_PP_EARLY_BAR
#define HF_H_PP_EARLY_BAR 0  // ping-pong K loop: 1 = a half passes the barrier its partner waits at BEFORE the MFMAs of its last tap (A/B builds)
#endif
#ifndef HF_H_PP_EARLY_X
#define HF_H_PP_EARLY_X 0  // ping-pong K loop: activation copies of the next stage the FIRST half issues right after requesting its first fragments (under their LDS latency, after the end-of-stage barrier) instead of in its idle phase (A/B builds)
#endif
#ifndef HF_H_FETCH_ORDER
#define HF_H_FETCH_ORDER 0  // 1 = a tap's fragment reads in the order its MFMAs consume them (a-hi, b-hi, b-lo, a-lo) instead of a-hi a-lo b-hi b-lo
#endif
#ifndef HF_H_PP_PREFETCH
#define HF_H_PP_PREFETCH 1  // ping-pong K loop: the half that computes second fetches its first tap's fragments BEFORE the role-swap barrier (0 = after: A/B builds)
#endif
#ifndef HF_H_ABLATE
#define HF_H_ABLATE 0  // timing experiments only: 1 no activation loads, 2 no epilogue stores, 4 no weight DMA, 8 no activation DMA
#endif

#ifdef HF_H_TRACE
// kernel-development build only: wave-level timeline of block 0 (s_memtime at pipeline points)
__device__ unsigned long long hf_trace_buf[8 * 512];
#define HF_TRACE_POINT(id)                                                                    \
  do {                                                                                         \
    if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && wave < 8 && trace_n < 510) {                    \
      hf_trace_buf[wave * 512 + trace_n++] = ((unsigned long long)(id) << 56) | (__builtin_readcyclecounter() & 0xffffffffffffffULL); \
    }                                                                                          \
  } while (0)
extern "C" int hf_debug_read_trace(unsigned long long *host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(hf_trace_buf), sizeof(unsigned long long) * n);
}
#else
#define HF_TRACE_POINT(id) ((void)0)
#endif

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
constexpr int KH = 16;  // input channels per stage = K of one MFMA

// Scheduling pattern of one tap-step (HF_H_ILV): NR times (one MFMA, one LDS read), then the remaining MFMAs
#ifndef HF_H_ILV_MODE
#define HF_H_ILV_MODE 1  // 1: (MFMA, read) x NR, MFMA x rest; 2: the reads behind the LAST NR MFMAs; 3: two reads behind each of the first NR / 2 MFMAs (A/B builds)
#endif
template <int NR, int NM>
__device__ __forceinline__ void hf_interleave() {
  if constexpr (HF_H_ILV_MODE == 2 && NM > NR) __builtin_amdgcn_sched_group_barrier(0x008, NM - NR, 0);
  if constexpr (HF_H_ILV_MODE == 3) {
#pragma unroll
    for (int k = 0; k < NR / 2; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
    if constexpr (NR & 1) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    if constexpr (NM > (NR + 1) / 2) __builtin_amdgcn_sched_group_barrier(0x008, NM - (NR + 1) / 2, 0);
  } else {
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
    }
    if constexpr (HF_H_ILV_MODE != 2 && NM > NR) __builtin_amdgcn_sched_group_barrier(0x008, NM - NR, 0);
  }
}

// Halo pixels the LDS activation tile is sized for: PT-pixel tiles of 32..TWMAX-pixel rows, and
// (UP) the 2-row / 2-column rim tiles of make_geom(one_image).  Wide tiles (TWMAX 128) turn the
// 128-byte row segments of a 32-pixel-wide tile into 512-byte ones: the high-resolution layers
// are HBM-bound and DRAM row locality decides their bandwidth.
// FUSE: overlapping th x 32 tiles only, no rim families (see below) - the four-wave form's two stage buffers then fit
// twice into a CU's 160 KB (2 x 37.4 KB per block instead of 2 x 43.2 KB).
template <int PT, bool UP, int TWMAX, bool FUSE = false>
constexpr int halo_pixels_max() {
  constexpr int h = UP ? 1 : 2;
  constexpr int narrow = (PT / 32 + h) * (32 + h), wide = (PT / TWMAX + h) * (TWMAX + h);
  constexpr int main_tile = wide > narrow ? wide : narrow;
  constexpr int rim = PT >= 512 ? 4 : 2;  // rows (columns) of a rim tile: keeps its halo below the main tile's
  constexpr int rim_tile = (rim + 1) * (PT / rim + 1);
  return (UP && !FUSE && rim_tile > main_tile) ? rim_tile : main_tile;
}

// UP: the transposed (stride 2) conv of the upsampling StyledConv, as in modconv.hip: 4 output
// phases (pr,pc) = (ky&1, kx&1) per input position (Y,X) of the (h+1)x(w+1) phase domain, each
// tap feeding exactly one phase from x[Y - (ky==2), X - (kx==2)]; no zero-insertion flops.
// PRE: the activations arrive pre-split and K-blocked (ConvParams::xh/xl): the stage's halo tile
// is fetched by LDS-DMA like the weights - no per-element loads, no conversion, no s.
// FUSE (UP only): the 4x4 blur + noise + bias + leaky ReLU of the upsampling StyledConv (model.py:263,
// :337-343) in the epilogue - the (2h+1)^2 intermediate is never written.  Tiles of th x 32 phase-domain
// positions OVERLAP by two positions (origin -1, stride th-2 / 30): a block computes the transposed conv on
// its whole tile and emits the blurred output of the interior positions, whose 4-tap windows it then holds
// completely (T is zero outside [0,2h] x [0,2w], which the zero-filled halo reproduces by itself, so there
// are no rim tile families).  Horizontal taps come from the neighbouring lanes (shuffles), vertical taps
// from the wave's other row or - across waves - through the LDS stage buffer that is free during the
// epilogue, four channels per lane at a time.  The blur kernel must be separable (rank 1).
// Order in which a chunk's 9 taps are processed.  Same-res: natural order, every tap has its own activation
// fragment.  UP: a tap reads the input position (Y - (ky==2), X - (kx==2)), so only FOUR distinct activation
// fragments exist per chunk; the taps are grouped by fragment - {0,1,3,4} {2,5} {6,7} {8} - and a fragment is
// fetched from LDS once per group instead of once per tap (8*PG instead of 18*PG ds_read_b128 per chunk: with
// 32-channel-wide wave tiles the fragment reads alone saturated the LDS port at the MFMA rate).
#ifndef HF_H_TAP_NATURAL
#define HF_H_TAP_NATURAL 0  // experiments: 1 = natural tap order for UP as well (one fragment fetch per tap)
#endif
template <bool UP_>
__host__ __device__ constexpr int tap_at(int i) {
  constexpr bool UP = UP_ && !HF_H_TAP_NATURAL;
  return !UP ? i : (i == 2 ? 3 : i == 3 ? 4 : i == 4 ? 2 : i);  // 0 1 3 4 2 5 6 7 8
}
template <bool UP_>
__host__ __device__ constexpr int tap_group(int i) {  // index of the activation fragment of position i
  constexpr bool UP = UP_ && !HF_H_TAP_NATURAL;
  return !UP ? i : (i < 4 ? 0 : i < 6 ? 1 : i < 8 ? 2 : 3);
}
template <bool UP_>
__host__ __device__ constexpr int group_first(int g) {  // first position of group g (9 = none)
  constexpr bool UP = UP_ && !HF_H_TAP_NATURAL;
  return !UP ? g : (g == 0 ? 0 : g == 1 ? 4 : g == 2 ? 6 : g == 3 ? 8 : 9);
}

// (second launch bound = waves per SIMD the register allocation must admit: a 16-wave block is four per SIMD, 128 registers)
template <int NTERMS, int CT_TILES, int PG, int WAVES_CO, int WAVES_PX, bool MOD, bool UP, int TWMAX, bool PRE, bool FUSE = false>
__global__ __launch_bounds__(64 * WAVES_CO * WAVES_PX, (WAVES_CO * WAVES_PX >= 16 ? 4 : 2)) void conv_mfma_h(const ConvParams P,
                                                                          const _Float16 *__restrict__ wth,
                                                                          const _Float16 *__restrict__ wtl) {
  constexpr int NW = WAVES_CO * WAVES_PX;
  constexpr int NT = 64 * NW;
  constexpr int CT = 32 * CT_TILES * WAVES_CO;
  constexpr int PT = 32 * PG * WAVES_PX;  // pixels (UP: phase-domain positions) per tile
  constexpr int NPH = UP ? 4 : 1;
  constexpr int HALO = UP ? 1 : 2;
  constexpr int NPIX = halo_pixels_max<PT, UP, TWMAX, FUSE>();
  constexpr int NPART = (NTERMS == 3) ? 2 : 1;          // hi (+ lo)
  constexpr int W_UNITS = 9 * 2 * CT;                   // 16-byte units of one weight part per stage
  constexpr int X_UNITS = 2 * NPIX;                     // 16-byte units of one activation part per stage
  // FUSE: the epilogue's vertical exchange (NW * 6 slots of 64 lanes x 16 B) lives in the free stage buffer - with plain
  // fp16 operands (one part) a stage is smaller than that, the buffer is sized for the exchange
  // (one row per wave, PG == 1: the row's four phases once - 4 slots; two rows per wave: the upper row's four + the lower row's two)
  constexpr int XSLOTS = (PG == 1) ? 4 : 6;
  constexpr int BUF_UNITS = (FUSE && NPART * (W_UNITS + X_UNITS) < NW * XSLOTS * 64) ? NW * XSLOTS * 64 : NPART * (W_UNITS + X_UNITS);
  constexpr int N_WPIECE = NPART * W_UNITS / 64;        // 1 KiB DMA pieces per stage
  // PP: the ping-pong K loop (below); PPR: its copy roles - activations by waves NW/2.., weights by waves 0..NW/2-1
  constexpr bool PP = PRE && NW == 8 && HF_H_PINGPONG;
  constexpr bool PPR = PP && HF_H_PP_ROLES;
  constexpr int WNW = PPR ? NW / 2 : NW;                // waves that issue weight copies
  constexpr int XNT = PPR ? NT / 2 : NT;                // threads that issue activation copies
  constexpr int ND = (N_WPIECE + WNW - 1) / WNW;        // per issuing wave
  // weight DMAs: all in the first three tap-steps when activations are staged through registers
  // (their loads must be issued first thing); one per tap-step when everything is DMA (PRE) -
  // spreading the arrivals over the chunk measured +1..3 %
  constexpr int DMA_PER_STEP = PRE ? 1 : (ND + 2) / 3;
  constexpr int XE = (X_UNITS + XNT - 1) / XNT;         // (pixel, kgroup) items per issuing thread per stage
  static_assert(W_UNITS % 64 == 0, "weight part must be whole 1 KiB pieces");
  static_assert(!FUSE || (UP && CT_TILES == 1 && WAVES_CO == 1 && TWMAX == 32), "FUSE: 32 co x (PG*WAVES_PX rows of 32 positions)");
  static_assert(!FUSE || BUF_UNITS * 16 >= NW * XSLOTS * 64 * 16, "FUSE: the exchange of one channel quad must fit a stage buffer");
  static_assert(!PRE || (ND <= 9 && XE <= 8), "PRE issue schedule: one weight DMA per tap-step, activations in odd steps (or every step)");

  HF_DYN_LDS;
  half8 *lds = reinterpret_cast<half8 *>(hf_dyn_lds);               // [2][BUF_UNITS] 16-byte units
  // buffer layout (units): [W hi][W lo][X hi][X lo]
  constexpr int OFF_WL = W_UNITS, OFF_XH = NPART * W_UNITS, OFF_XL = NPART * W_UNITS + X_UNITS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int li = lane & 31;
  const int lh = lane >> 5;  // k group of the lane: input channels 8*lh .. 8*lh+7 of the stage
  const int xtid = PPR ? tid - NT / 2 : tid;  // index among the threads that stage activations (negative: none)
  const int wave_co = (wave / WAVES_PX) * (32 * CT_TILES);
  const int wave_pg = (wave % WAVES_PX) * PG;
  // grid = (tile walkers, cout tiles), or - ConvParams::swap_xy - (cout tiles, tile walkers): blocks are dispatched x-fastest, so
  // consecutive blocks then share an input tile and differ in the cout tile (block b runs on XCD b % 8: with 8 cout tiles every
  // XCD keeps ONE cout tile's weights in its L2 and the input tiles stream through) - see launch_h
  const int blk_x = P.swap_xy ? (int)blockIdx.y : (int)blockIdx.x, blk_y = P.swap_xy ? (int)blockIdx.x : (int)blockIdx.y;
  const int grid_x = P.swap_xy ? (int)gridDim.y : (int)gridDim.x, grid_y = P.swap_xy ? (int)gridDim.x : (int)gridDim.y;
  const int co0 = blk_y * CT;

  const long long plane = (long long)P.h * P.w;
  const int iplane = (int)plane;
  const int nchunks = P.cin / KH;
  float *sl_base = reinterpret_cast<float *>(lds + 2 * BUF_UNITS);  // [2][cin]: s of the current / next image
  // epilogue parameters of the block's CT output channels, per image slot: [2][3][CT] = d, bias,
  // s_next (modulation of the consumer, split output only).
  // In LDS so that the epilogue issues NO vector-memory loads: on gfx9 loads and stores share
  // vmcnt, every load that follows a store waits for that store's acknowledgement (measured:
  // the epilogue's second pixel group waited ~12k cycles behind the first group's stores).
  float *ep_base = sl_base + 2 * ((P.cin + 3) & ~3);
  float *rgbw_base = ep_base + 2 * 3 * CT;  // fused ToRGB: [2 slots][3][CT] = rgb_w[co][c] * rgb_s[b][co]

  // ---- tiles: the block walks tiles blockIdx.x, +gridDim.x, ... of its cout tile as ONE
  // pipeline - the first stage of the next tile is prefetched during the last stage of the
  // current one and the epilogue stores drain under the next tile's first stage ----
  struct Tile {
    int gi, b0, ty0, tx0;
  };
  // constant indices only: a runtime index into the by-value kernel argument would make the
  // compiler copy the whole struct to scratch memory
  const TileGeom G0 = P.g[0], G1 = P.g[1], G2 = P.g[2];
  auto geom = [&](int gi) {
    TileGeom G;
#define HF_PICK(f) G.f = gi == 0 ? G0.f : (gi == 1 ? G1.f : G2.f)
    HF_PICK(y0); HF_PICK(x0); HF_PICK(dh); HF_PICK(dw); HF_PICK(lg_tw); HF_PICK(lg_th); HF_PICK(lg_nb);
    HF_PICK(tiles_x); HF_PICK(tiles_y); HF_PICK(tiles_b); HF_PICK(first_block);
#undef HF_PICK
    return G;
  };
  auto locate = [&](int t) {  // flat tile index -> family, image, first pixel
    Tile T;
    T.gi = 0;
    if (P.n_geom > 1 && t >= G1.first_block) T.gi = 1;
    if (P.n_geom > 2 && t >= G2.first_block) T.gi = 2;
    const TileGeom G = geom(T.gi);
    // (integer divisions: a reciprocal-multiply form of these and of the halo-item division below measured
    // no gain on the same-resolution kernels and 5-9 % slower fused upsampling kernels)
    int r = t - G.first_block;
    const int tx = r % G.tiles_x;
    r /= G.tiles_x;
    const int ty = r % G.tiles_y;
    T.b0 = r / G.tiles_y;  // one image per tile
    T.ty0 = G.y0 + (ty << G.lg_th);
    T.tx0 = G.x0 + (tx << G.lg_tw);
    if (FUSE) {  // overlapping tiles: origin -1, interior (th-2) x (tw-2) positions per tile
      T.ty0 = ty * ((1 << G.lg_th) - 2) - 1;
      T.tx0 = tx * ((1 << G.lg_tw) - 2) - 1;
    }
    return T;
  };
  // staging items of this thread: (halo pixel, kgroup) -> offset inside a channel plane
  // (-1: outside the image, zero fill; -2: no item); the halo tile starts at (ty0-1, tx0-1)
  constexpr int NSRC = XE;
  auto locate_items = [&](const Tile &T, int (&src)[NSRC]) {
    const TileGeom G = geom(T.gi);
    const int wp = (1 << G.lg_tw) + HALO, xs = ((1 << G.lg_th) + HALO) * wp;
#pragma unroll
    for (int e = 0; e < XE; ++e) {
      const int i = xtid + e * XNT;
      const int kg = i / NPIX, pix = i - kg * NPIX;
      src[e] = -2;
      if (i >= 0 && i < X_UNITS && pix < xs) {
        const int hy = pix / wp, hx = pix - hy * wp;
        const int ys = T.ty0 + hy - 1, xc = T.tx0 + hx - 1;
        src[e] = (ys >= 0 && ys < P.h && xc >= 0 && xc < P.w) ? ys * P.w + xc : -1;
      }
    }
  };
  // 2^-k of the weights' power-of-two pre-scale (trailer of wt_hi, see split_weights): folded into d
  const float w_unscale = *reinterpret_cast<const float *>(wth + 9LL * P.cin * P.cout);
  // Straight-line epilogue for the generator's standard StyledConv tail (noise + bias + leaky ReLU, * scale):
  // lrelu is positively homogeneous, so with 0 <= alpha <= 1 and scale > 0
  //   scale * lrelu(acc*d + n + b) = max(o, alpha*o),  o = acc*(d*scale) + (n*scale + b*scale)
  // - d*scale and b*scale are folded into the per-image epilogue table below.  Anything else (no bias, other
  // activations) takes the general epilogue with its run-time switches.
  const bool fast_ep = !UP && P.bias && P.act == ACT_LRELU && P.alpha >= 0.0f && P.alpha <= 1.0f && P.scale > 0.0f;
  const float ep_fold = (FUSE || fast_ep) ? P.scale : 1.0f;
  auto load_s = [&](int b, int slot) {
    float *dst = sl_base + slot * P.cin;
    if (MOD)
      for (int i = tid; i < P.cin; i += NT) dst[i] = P.s[(long long)b * P.s_bstride + i];
    float *ep = ep_base + slot * 3 * CT;
    for (int i = tid; i < CT; i += NT) {
      ep[i] = (P.d ? P.d[(long long)b * P.d_bstride + co0 + i] : 1.0f) * w_unscale * ep_fold;
      ep[CT + i] = P.bias ? P.bias[co0 + i] * ep_fold : 0.0f;
      ep[2 * CT + i] = ((!UP || FUSE) && P.oh && P.s_next) ? P.s_next[(long long)b * P.cout + co0 + i] : 1.0f;
    }
    if (!UP && P.rgb_out) {
      float *rw = rgbw_base + slot * 3 * CT;
      for (int i = tid; i < CT; i += NT) {
        const float sv = P.rgb_s[(long long)b * P.cout + co0 + i];
#pragma unroll
        for (int c = 0; c < 3; ++c) rw[c * CT + i] = P.rgb_w[(co0 + i) * 3 + c] * sv;
      }
    }
  };

  // Unmodulated / pre-split kernels (no s): the NEXT image's tables in two steps - the loads at the head of the tile into
  // registers, the LDS writes after the tile's K loop (HF_H_LATE_TABLES).  With the strided walk a block of a 512^2 layer
  // changes image on every second tile, one of a 256^2 layer on every tile; load_s at the tile head cost a block-wide barrier
  // and the exposed latency of its global loads in wave 0 - 5.9 k of the 83 k cycles of a 64 -> 64 @512^2 tile
  // (profiles/r06af_trace_same_res_64ch.txt).  No barrier is needed: the slot written after tile i's K loop was last read in
  // tile i-1's epilogue, which every wave has left (it passed tile i's stage barriers); readers (tile i+1's epilogue) are
  // behind tile i+1's stage barriers.  Same values, same bits.
  constexpr bool LATE_TABLES = !MOD && HF_H_LATE_TABLES;
  float pend_d = 1.0f, pend_b = 0.0f, pend_sn = 1.0f, pend_rs = 0.0f, pend_rw[3] = {0.0f, 0.0f, 0.0f};
  bool pend = false;
  auto load_tables_issue = [&](int b) {
    if (tid < CT) {
      pend_d = P.d ? P.d[(long long)b * P.d_bstride + co0 + tid] : 1.0f;
      pend_b = P.bias ? P.bias[co0 + tid] : 0.0f;
      pend_sn = ((!UP || FUSE) && P.oh && P.s_next) ? P.s_next[(long long)b * P.cout + co0 + tid] : 1.0f;
      if (!UP && P.rgb_out) {
        pend_rs = P.rgb_s[(long long)b * P.cout + co0 + tid];
#pragma unroll
        for (int c = 0; c < 3; ++c) pend_rw[c] = P.rgb_w[(co0 + tid) * 3 + c];
      }
    }
  };
  auto load_tables_commit = [&](int slot) {
    if (tid < CT) {
      float *ep = ep_base + slot * 3 * CT;
      ep[tid] = pend_d * w_unscale * ep_fold;
      ep[CT + tid] = P.bias ? pend_b * ep_fold : 0.0f;
      ep[2 * CT + tid] = pend_sn;
      if (!UP && P.rgb_out) {
        float *rw = rgbw_base + slot * 3 * CT;
#pragma unroll
        for (int c = 0; c < 3; ++c) rw[c * CT + tid] = pend_rw[c] * pend_rs;
      }
    }
  };

  // prefetch source (tile whose stage is being staged): items, image base, s
  int e_src[XE];
  const float *xb;
  int sl_off;  // offset of the prefetch source's s inside sl_base (an index, not a pointer: LDS
               // pointers that get selected at run time turn into generic pointers + aperture checks)
  int sl_slot = 0;

  // weight stage of `chunk`: uniform base + per-lane byte offset ((tap*2+kg)*cout + co0 + col)*16
  const unsigned lds_addr0 = hf_lds_addr(lds);
  auto dma_piece = [&](int i, int chunk, int bufsel) {
    const int pc = wave + i * WNW;
    if (wave < WNW && pc < N_WPIECE && !(HF_H_ABLATE & 4)) {
      const int part = pc / (W_UNITS / 64), q = pc % (W_UNITS / 64);
      const int u = q * 64 + lane;            // unit inside the part: (tap*2 + kg)*CT + co
      const int row = u / CT, col = u % CT;   // row = tap*2 + kg
      int off = (row * P.cout + co0 + col) * 16;
      HF_OPAQUE_I32(off);
      const _Float16 *src = (part ? wtl : wth) + (long long)chunk * 18 * P.cout * 8;
      hf_glds16_raw_s(src, (unsigned)off, lds_addr0 + (unsigned)(bufsel * BUF_UNITS + part * W_UNITS + q * 64) * 16u);
    }
  };

  // PRE: one (halo pixel, kgroup) unit per lane and item straight into LDS; halo pixels outside
  // the image are masked out of the DMA and zero-filled by the same lane
  const char *xh_b = nullptr, *xl_b = nullptr;  // image base of the pre-split tensors (uniform)
  auto dma_x = [&](int e, int chunk, int bufsel) {
    const int i = xtid + e * XNT;
    const int kg = i / NPIX;
    if (HF_H_ABLATE & 8) return;  // timing experiments: no activation DMA
    if (PPR && wave < NW / 2) return;  // (uniform per wave)
    const bool inside = e_src[e] >= 0;
    int off = inside ? (kg * iplane + e_src[e]) * 16 : 0;
    HF_OPAQUE_I32(off);
    const long long cofs = (long long)chunk * 2 * plane * 16;  // 2 channel blocks per stage
    const unsigned dst = lds_addr0 + (unsigned)(bufsel * BUF_UNITS + OFF_XH + (i - lane)) * 16u;
    hf_glds16_raw_s_if(inside, xh_b + cofs, (unsigned)off, dst);
    if (NTERMS == 3) hf_glds16_raw_s_if(inside, xl_b + cofs, (unsigned)off, dst + (unsigned)X_UNITS * 16u);
    if (e_src[e] == -1) {
      half8 z;
#pragma unroll
      for (int k = 0; k < 8; ++k) z[k] = (_Float16)0.0f;
      half8 *buf = lds + bufsel * BUF_UNITS;
      buf[OFF_XH + i] = z;
      if (NTERMS == 3) buf[OFF_XL + i] = z;
    }
  };

  float xr[XE][8];  // raw activations of the next stage, in flight for most of a chunk
  // uniform chunk base + per-lane offset; halo items outside the image load element 0 of the
  // chunk instead (unconditional loads: no branches in the pipeline) and are zeroed at conversion
  auto load_item = [&](int e, int chunk) {
    const char *xc = reinterpret_cast<const char *>(xb + (long long)chunk * KH * plane);  // uniform
    const int kg = (tid + e * NT) / NPIX;
    int off = (e_src[e] >= 0) ? kg * 8 * iplane + e_src[e] : 0;
    HF_OPAQUE_I32(off);  // addresses recomputed per chunk, not kept live as 8 hoisted 64-bit pairs
#pragma unroll
    for (int k = 0; k < 8; ++k)
      xr[e][k] = (HF_H_ABLATE & 1) ? 1.0f : *reinterpret_cast<const float *>(xc + (unsigned)((off + k * iplane) * 4));
  };
  auto convert_item = [&](int e, int chunk, half8 *buf) {
    if (e_src[e] == -2) return;
    const int i = tid + e * NT;
    const int kg = i / NPIX;
    half8 hi, lo;
    bool ovf = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float v = (e_src[e] >= 0) ? xr[e][k] : 0.0f;
      if (MOD) v *= sl_base[sl_off + chunk * KH + kg * 8 + k];
      _Float16 hv, lv;
      hf_split_f16(v, hv, lv, ovf);  // saturating; see hf_common.h
      hi[k] = hv;
      lo[k] = lv;
    }
    hf_note_overflow(ovf);
    buf[OFF_XH + i] = hi;
    if (NTERMS == 3) buf[OFF_XL + i] = lo;
  };

  f32x16 acc[NPH][CT_TILES][PG];
  auto zero_acc = [&]() {
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
      for (int ct = 0; ct < CT_TILES; ++ct)
#pragma unroll
        for (int g = 0; g < PG; ++g)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[ph][ct][g][r] = 0.0f;
  };
  zero_acc();

  int t_cur = blk_x;
  Tile cur = locate(t_cur);
  locate_items(cur, e_src);
  xb = P.x + (long long)cur.b0 * P.cin * plane;
  if (PRE) {
    xh_b = static_cast<const char *>(P.xh) + (long long)cur.b0 * (P.cin / 8) * plane * 16;
    xl_b = static_cast<const char *>(P.xl) + (long long)cur.b0 * (P.cin / 8) * plane * 16;
  }
  sl_off = 0;
  load_s(cur.b0, 0);
  int ep_slot = 0;  // slot of the tile being computed (its epilogue)
  __syncthreads();  // s visible
  // prologue: stage 0 of the first tile
#pragma unroll
  for (int i = 0; i < ND; ++i) dma_piece(i, 0, 0);
  if (PRE) {
#pragma unroll
    for (int e = 0; e < XE; ++e) dma_x(e, 0, 0);
  } else {
#pragma unroll
    for (int e = 0; e < XE; ++e) load_item(e, 0);
#pragma unroll
    for (int e = 0; e < XE; ++e) convert_item(e, 0, lds);
  }
  HF_H_BARRIER();

  int trace_n = 0;  // HF_H_TRACE builds: events recorded by this wave
  (void)trace_n;
  // Epilogue of one tile.  MFMA D layout: row (= co) = (r&3) + 8*(r>>2) + 4*lh, col (= pixel) = li:
  // the lane's 4 channels of register group q = r>>2 are consecutive -> one ds_read_b128 each of
  // d and bias.  Same-res: v = lrelu(acc*d + noise_w*noise + bias)*scale (bias NULL: v = acc*d);
  // UP: v = acc*d of the 4 phases into the (2h+1) x pitch intermediate, phases (pr,0),(pr,1) as
  // one 8-byte store.  Nothing but stores goes to vector memory here.
  const float nw_ = (!UP && P.noise) ? P.noise_w[0] : 0.0f;
  auto epilogue = [&](const TileGeom &G, const Tile &T, int slot, const float (&nz)[PG]) {
    // opaque per tile: otherwise the per-register channel offsets (64-bit, one per accumulator
    // register) are hoisted out of the tile loop and live - spilled - through it
    int co_w = wave_co, li_o = li, lh_o = lh;
    HF_OPAQUE_I32(co_w);
    HF_OPAQUE_I32(li_o);
    HF_OPAQUE_I32(lh_o);
    const int tw = 1 << G.lg_tw, th = 1 << G.lg_th;
    const float *ep = ep_base + slot * 3 * CT;
    // uniform base of the block's CT output planes of this image + 32-bit per-lane byte offsets
    // (the launcher checks CT * plane bytes < 4 GiB): one add per store instead of 64-bit math
    const unsigned oplane4 = (unsigned)(P.out_h * P.out_w) * 4u;
    char *ob0 = reinterpret_cast<char *>(P.out + ((long long)T.b0 * P.cout + co0) * ((long long)P.out_h * P.out_w));
#pragma unroll
    for (int g = 0; g < PG; ++g) {
      const int p = (wave_pg + g) * 32 + li_o;
      const int Y = T.ty0 + ((p >> G.lg_tw) & (th - 1)), X = T.tx0 + (p & (tw - 1));
      const bool pv = (Y < G.y0 + G.dh) && (X < G.x0 + G.dw);
      const float nzv = nw_ * nz[g];
      float rgb[3] = {0.0f, 0.0f, 0.0f};
      if (pv) {
      const unsigned pix4 = UP ? (unsigned)(2 * Y * P.out_w + 2 * X) * 4u : (unsigned)(Y * P.out_w + X) * 4u;
#pragma unroll
      for (int ct = 0; ct < CT_TILES; ++ct)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c4 = co_w + ct * 32 + 8 * q + 4 * lh_o;  // first of the lane's 4 channels (tile-relative)
          const float4 dm = *reinterpret_cast<const float4 *>(ep + c4);
          const float4 bs = *reinterpret_cast<const float4 *>(ep + CT + c4);
          const float dmv[4] = {dm.x, dm.y, dm.z, dm.w}, bsv[4] = {bs.x, bs.y, bs.z, bs.w};
          unsigned off = (unsigned)c4 * oplane4 + pix4;
#pragma unroll
          for (int k = 0; k < 4; ++k, off += oplane4) {
            const int r = 4 * q + k;
            if (UP) {
#pragma unroll
              for (int pr = 0; pr < 2; ++pr) {
                if (2 * Y + pr >= P.out_h) continue;
                float *qo = reinterpret_cast<float *>(ob0 + (off + (unsigned)(pr * P.out_w) * 4u));
                const float v0 = acc[2 * pr][ct][g][r] * dmv[k], v1 = acc[(UP ? 2 * pr + 1 : 0)][ct][g][r] * dmv[k];
                if (2 * X + 1 < P.out_wv) {
                  f32x2u pair;
                  pair.x = v0;
                  pair.y = v1;
                  *reinterpret_cast<f32x2u *>(qo) = pair;
                } else {
                  qo[0] = v0;
                }
              }
            } else {
              float v = acc[0][ct][g][r] * dmv[k];
              if (P.bias) v = apply_act(v + nzv + bsv[k], P.act, P.alpha, P.scale, 0.0f);
              if (P.out) *reinterpret_cast<float *>(ob0 + off) = v;  // null: only the fused ToRGB consumes it
              acc[0][ct][g][r] = v;  // kept for the fused ToRGB / split output below
            }
          }
          if (!UP && P.oh) {
            // split output for a pre-split consumer: the lane's 4 channels are one half (lh) of the
            // 16-byte unit of pixel (Y, X), channel block (co0 + c4) / 8: two 8-byte stores; the two
            // half-waves together write 512 contiguous bytes per 32 pixels
            const float4 sn = *reinterpret_cast<const float4 *>(ep + 2 * CT + c4);
            const float snv[4] = {sn.x, sn.y, sn.z, sn.w};
            typedef _Float16 half4 __attribute__((ext_vector_type(4)));
            half4 h4, l4;
            bool ovf = false;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              _Float16 hv, lv;
              hf_split_f16(acc[0][ct][g][4 * q + k] * snv[k], hv, lv, ovf);
              h4[k] = hv;
              l4[k] = lv;
            }
            hf_note_overflow(ovf);
            const long long unit = (((long long)T.b0 * (P.cout >> 3) + ((co0 + c4) >> 3)) * P.out_h + Y) * P.out_w + X;
            *reinterpret_cast<half4 *>(static_cast<char *>(P.oh) + unit * 16 + ((co0 + c4) & 4) * 2) = h4;
            if (P.ol) *reinterpret_cast<half4 *>(static_cast<char *>(P.ol) + unit * 16 + ((co0 + c4) & 4) * 2) = l4;
          }
        }
      }  // pv
      if (!UP && P.rgb_out) {
        // fused ToRGB (model.py:356-362 without bias / skip): the wave holds all CT = cout channels
        // of its pixels (WAVES_CO == 1, one cout tile): 3 dot products over the lane's 16*CT_TILES
        // values, then the two half-waves (channels +4) are added and the low half stores.
        // The shuffle runs for every lane (pixels outside the image contribute to nothing).
        const float *rw = rgbw_base + slot * 3 * CT;
        if (pv) {
#pragma unroll
          for (int ct = 0; ct < CT_TILES; ++ct)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int c4 = co_w + ct * 32 + 8 * q + 4 * lh_o;
#pragma unroll
              for (int c = 0; c < 3; ++c) {
                const float4 wv = *reinterpret_cast<const float4 *>(rw + c * CT + c4);
                rgb[c] = fmaf(acc[0][ct][g][4 * q + 0], wv.x, rgb[c]);
                rgb[c] = fmaf(acc[0][ct][g][4 * q + 1], wv.y, rgb[c]);
                rgb[c] = fmaf(acc[0][ct][g][4 * q + 2], wv.z, rgb[c]);
                rgb[c] = fmaf(acc[0][ct][g][4 * q + 3], wv.w, rgb[c]);
              }
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          rgb[c] += __shfl_xor(rgb[c], 32, 64);
          if (pv && lh_o == 0)
            P.rgb_out[((long long)T.b0 * 3 + c) * ((long long)P.out_h * P.out_w) + (long long)Y * P.out_w + X] = rgb[c];
        }
      }
    }
  };

  // ---- same-resolution fast epilogue (fast_ep): the accumulators are only read and the run-time switches (which
  // outputs exist) are tested once per channel quad, not per element - the general epilogue spent ~9k cycles per
  // 512-pixel tile of the 1024^2 layer on scalar branches, a third of the tile.  (One body with uniform branches:
  // separate compile-time variants made LLVM hoist the common arithmetic above the dispatch and spill it.) ----
  auto epilogue_fast = [&](const TileGeom &G, const Tile &T, int slot, const float (&nz)[PG]) {
    const bool OUT = P.out != nullptr, SPLIT = P.oh != nullptr, RGB = P.rgb_out != nullptr;  // uniform
    int co_w = wave_co, li_o = li, lh_o = lh;
    HF_OPAQUE_I32(co_w);
    HF_OPAQUE_I32(li_o);
    HF_OPAQUE_I32(lh_o);
    const int tw = 1 << G.lg_tw, th = 1 << G.lg_th;
    const float *ep = ep_base + slot * 3 * CT;
    const float *rw = rgbw_base + slot * 3 * CT;
    const unsigned oplane4 = (unsigned)(P.out_h * P.out_w) * 4u;
    char *ob0 = OUT ? reinterpret_cast<char *>(P.out + ((long long)T.b0 * P.cout + co0) * ((long long)P.out_h * P.out_w)) : nullptr;
    // split output: uniform base of the block's CT / 8 channel-block planes of this image + 32-bit per-lane byte offsets (CT / 8
    // planes of 16-byte units: the launcher's CT * plane * 4 < 4 GiB bound covers them) - the 64-bit unit arithmetic per channel
    // quad was ~18 VALU instructions, five of them quarter-rate multiplies, in an epilogue that is bound by VALU issue
    const unsigned uplane16 = (unsigned)(P.out_h * P.out_w) * 16u;
    const long long ubase = ((long long)T.b0 * (P.cout >> 3) + (co0 >> 3)) * ((long long)P.out_h * P.out_w) * 16;
    char *oh_t = SPLIT ? static_cast<char *>(P.oh) + ubase : nullptr;
    char *ol_t = (SPLIT && P.ol) ? static_cast<char *>(P.ol) + ubase : nullptr;
    const float nws = nw_ * P.scale;
    bool ovf_tile = false;
    // per pixel group: validity, offsets, noise term (a product of its own in every instantiation - no contraction into the add
    // below: the tile configurations must agree bit for bit)
    bool pvg[PG];
    unsigned pix4[PG], pix16[PG];
    int Yg[PG], Xg[PG];
    float nzv[PG], rgb[PG][3];
#pragma unroll
    for (int g = 0; g < PG; ++g) {
      const int p = (wave_pg + g) * 32 + li_o;
      Yg[g] = T.ty0 + ((p >> G.lg_tw) & (th - 1));
      Xg[g] = T.tx0 + (p & (tw - 1));
      pvg[g] = (Yg[g] < G.y0 + G.dh) && (Xg[g] < G.x0 + G.dw);
      pix4[g] = (unsigned)(Yg[g] * P.out_w + Xg[g]) * 4u;
      pix16[g] = (unsigned)(Yg[g] * P.out_w + Xg[g]) * 16u + (unsigned)lh_o * 8u;
      nzv[g] = nws * nz[g];
      HF_OPAQUE_F32(nzv[g]);
      rgb[g][0] = rgb[g][1] = rgb[g][2] = 0.0f;
    }
    // channel quad outermost: its per-channel constants are fetched from LDS once, together, for both pixel groups (they were
    // fetched per group and consumed one by one: up to six exposed LDS latencies per quad and group).  Every lane runs the
    // arithmetic (the split's range vote is wave-wide; lanes outside the image hold finite sums of zero padding), lanes of
    // valid pixels store.  Per element the operations and their order are unchanged: same bits.
#pragma unroll
    for (int ct = 0; ct < CT_TILES; ++ct)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c4 = co_w + ct * 32 + 8 * q + 4 * lh_o;  // first of the lane's 4 channels (tile-relative)
        const float4 dm = *reinterpret_cast<const float4 *>(ep + c4);
        const float4 bs = *reinterpret_cast<const float4 *>(ep + CT + c4);
        float4 sn = make_float4(1.0f, 1.0f, 1.0f, 1.0f), wv[3];
        if (SPLIT) sn = *reinterpret_cast<const float4 *>(ep + 2 * CT + c4);
        if (RGB) {
#pragma unroll
          for (int c = 0; c < 3; ++c) wv[c] = *reinterpret_cast<const float4 *>(rw + c * CT + c4);
        }
        const float dmv[4] = {dm.x, dm.y, dm.z, dm.w}, bsv[4] = {bs.x, bs.y, bs.z, bs.w};
        const unsigned cofs4 = (unsigned)c4 * oplane4;                       // fp32 planes of the lane's channels
        const unsigned cofs16 = (unsigned)((c4 - 4 * lh_o) >> 3) * uplane16;  // the 16-byte-unit plane of channel block c4 / 8
#pragma unroll
        for (int g = 0; g < PG; ++g) {
          float v[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float o = fmaf(acc[0][ct][g][4 * q + k], dmv[k], nzv[g] + bsv[k]);
            v[k] = fmaxf(o, o * P.alpha);
          }
          if (OUT && pvg[g]) {
            unsigned off = cofs4 + pix4[g];
#pragma unroll
            for (int k = 0; k < 4; ++k, off += oplane4) *reinterpret_cast<float *>(ob0 + off) = v[k];
          }
          if (SPLIT) {  // the lane's 4 channels = one half (lh) of the 16-byte unit of pixel (Y, X), channel block (co0+c4)/8
            const float vs[4] = {v[0] * sn.x, v[1] * sn.y, v[2] * sn.z, v[3] * sn.w};
            hf_half4 h4, l4;
            bool ovf = false;
            hf_split4_f16(vs, h4, l4, ovf);  // one range vote per four values, packed conversions (2.5 VALU issues per element)
            if (pvg[g]) {
              ovf_tile = ovf_tile || ovf;
              const unsigned off = cofs16 + pix16[g];
              *reinterpret_cast<hf_half4 *>(oh_t + off) = h4;
              if (ol_t) *reinterpret_cast<hf_half4 *>(ol_t + off) = l4;
            }
          }
          if (RGB) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
              rgb[g][c] = fmaf(v[3], wv[c].w, fmaf(v[2], wv[c].z, fmaf(v[1], wv[c].y, fmaf(v[0], wv[c].x, rgb[g][c]))));
          }
        }
      }
    if (RGB) {
      // the two half-waves hold the other channels of the same pixels: add, the low half stores.  A wave covers
      // 32*CT_TILES of the cout channels: its sum goes to slab (blockIdx.y*WAVES_CO + co-wave) of the
      // [B][slabs*3][H][W] raw tensor; ToRGB's finishing pass adds the slabs in a fixed order (deterministic)
      const int slabs = grid_y * WAVES_CO, slab = blk_y * WAVES_CO + wave / WAVES_PX;
#pragma unroll
      for (int g = 0; g < PG; ++g)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float sum = rgb[g][c] + __shfl_xor(rgb[g][c], 32, 64);
          if (pvg[g] && lh_o == 0)
            P.rgb_out[(((long long)T.b0 * slabs + slab) * 3 + c) * ((long long)P.out_h * P.out_w) + (long long)Yg[g] * P.out_w + Xg[g]] = sum;
        }
    }
    if (SPLIT) hf_note_overflow(ovf_tile);
  };

  // ---- FUSE: blur + noise + bias + lrelu (+ split) epilogue of the transposed conv --------------------
  // acc[pr*2+pc][0][g][r] = T[2Y+pr][2X+pc] of position (Y, X) = (ty0 + wave_pg + g, tx0 + li), channel
  // co0 + (r&3) + 8*(r>>2) + 4*lh.  out[oy][ox] = sum_{r,j} ky[r] kx[j] T[oy-1+r][ox-1+j] with the FLIPPED
  // 1-D factors of the blur kernel (upfirdn2d is a true convolution, op/upfirdn2d.py:186).
  // 1-D factors of the rank-1 blur kernel, already flipped, as kernel arguments (scalar registers)
  const float kxf[4] = {P.blur_kx[0], P.blur_kx[1], P.blur_kx[2], P.blur_kx[3]};
  const float kyf[4] = {P.blur_ky[0], P.blur_ky[1], P.blur_ky[2], P.blur_ky[3]};
  // SPLIT (compile time): write the fp16 (hi, lo) split of s_next*out (P.oh / P.ol) or the fp32 tensor (P.out).
  // The launcher guarantees bias != NULL, act == leaky ReLU with 0 <= alpha <= 1 (so lrelu(v) = max(v, alpha*v))
  // and scale > 0 (positively homogeneous: folded into d, the noise weight and the bias).
  auto epilogue_fused = [&](auto SPLIT_T, const Tile &T, int slot, const float (&nz)[PG][4], int freebuf) {
    constexpr bool SPLIT = decltype(SPLIT_T)::value;
    bool ovf_tile = false;
    int li_o = li, lh_o = lh;
    HF_OPAQUE_I32(li_o);
    HF_OPAQUE_I32(lh_o);
    const float *ep = ep_base + slot * 3 * CT;
    const int wv = wave;  // WAVES_CO == 1: wave = pixel wave, tile rows wave*PG + g
    // 2. vertical pass per channel quad q (registers 4q..4q+3): the rows of the neighbouring WAVES travel
    //    through LDS (24 floats per lane and wave), the wave's own rows are in registers
    float *xch = reinterpret_cast<float *>(lds + freebuf * BUF_UNITS);  // [wave][24][64 lanes]
    const int row0 = wv * PG;  // tile row of g = 0
    const int Y0 = T.ty0 + row0, X = T.tx0 + li_o;
    const float nw_f = (P.noise ? P.noise_w[0] : 0.0f) * P.scale;
    const long long oplane = (long long)(2 * P.h) * (2 * P.w);
    // byte offset of output pixel (2*Y0, 2*X) inside a 16-byte-unit plane / a float plane (32-bit: <= 64 MiB planes)
    const int pix00 = (2 * Y0) * (2 * P.w) + 2 * X;
    const bool colv = li_o >= 1 && li_o <= 30 && X >= 0 && X < P.w;
    char *oh_b = SPLIT ? static_cast<char *>(P.oh) + ((long long)T.b0 * (P.cout >> 3) + (co0 >> 3)) * oplane * 16 : nullptr;
    char *ol_b = SPLIT ? static_cast<char *>(P.ol) + ((long long)T.b0 * (P.cout >> 3) + (co0 >> 3)) * oplane * 16 : nullptr;
    float *of_b = SPLIT ? nullptr : P.out + ((long long)T.b0 * P.cout + co0) * oplane;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      __builtin_amdgcn_sched_barrier(0);  // one channel quad at a time: keeps the live set small
      // horizontal pass of this quad: (T[.][2X], T[.][2X+1]) -> (H[.][2X], H[.][2X+1]) from the neighbouring lanes'
      // values; the accumulators are only READ (writing elements back into the 16-wide accumulator vectors makes
      // the compiler copy whole register tuples).  Lanes 0 / 31 of a half-wave hold tile-edge positions whose
      // results are never used (their neighbours belong to the other half).
      // (Round 4, measured and rejected again: the same arithmetic on channel pairs as ext_vector float2 - every multiply-add
      // a v_pk_fma_f32 / v_pk_mul_f32: 7413 -> 6322 VALU instructions in the kernel, but the eight broadcast tap pairs and
      // the paired temporaries cost 76 B more scratch with reloads INSIDE the K loop: 595 -> 670 us on the 1024^2 layer.)
      float H[4][PG][4];  // [pr*2+pc][g][k]
#pragma unroll
      for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int g = 0; g < PG; ++g)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float t0 = acc[2 * pr][0][g][4 * q + k], t1 = acc[2 * pr + 1][0][g][4 * q + k];
            const float l1 = hf_lane_up(t1), r0 = hf_lane_down(t0), r1 = hf_lane_down(t1);
            H[2 * pr][g][k] = fmaf(kxf[3], r0, fmaf(kxf[2], t1, fmaf(kxf[1], t0, kxf[0] * l1)));
            H[2 * pr + 1][g][k] = fmaf(kxf[3], r1, fmaf(kxf[2], r0, fmaf(kxf[1], t1, kxf[0] * t0)));
          }
      // publish (one 16-byte unit per lane and slot): slots 0..3 = the g = 0 row for the wave ABOVE (index pr*2+pc),
      // 4..5 = phase 1 of the last row for the wave BELOW (index pc); values are read back right where they are
      // used (nothing but the current output pixel's operands is live: the epilogue must not spill - a scratch
      // reload waits for every output store issued before it)
      HF_TRACE_POINT(30);  // fused epilogue: horizontal pass of the quad done
      float4 *xq = reinterpret_cast<float4 *>(xch);  // [wave][6][64 lanes]
      float4 *mine = xq + (wv * XSLOTS) * 64 + lane;
      if (!(HF_H_EPI_ABLATE & 4)) {
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) mine[ph * 64] = make_float4(H[ph][0][0], H[ph][0][1], H[ph][0][2], H[ph][0][3]);
        if (PG > 1) {
#pragma unroll
          for (int pc = 0; pc < 2; ++pc)
            mine[(4 + pc) * 64] = make_float4(H[2 + pc][PG - 1][0], H[2 + pc][PG - 1][1], H[2 + pc][PG - 1][2], H[2 + pc][PG - 1][3]);
        }
      }
      HF_TRACE_POINT(31);  // rows published, before the barrier
      if (!(HF_H_EPI_ABLATE & 4)) hf_barrier_lds();
      HF_TRACE_POINT(32);  // after the barrier
      const float4 *above = xq + (max(wv - 1, 0) * XSLOTS) * 64 + lane, *below = xq + (min(wv + 1, NW - 1) * XSLOTS) * 64 + lane;
      constexpr int ABOVE_P1 = (PG == 1) ? 2 : 4;  // slot of phase (1, pc) of the LAST row of the wave above
      const int c4 = 8 * q + 4 * lh_o;  // first of the lane's 4 channels (tile relative)
      // the lane's 4 channels are one half (lh) of the 16-byte unit of channel block q
      const long long cb_ofs = (long long)q * oplane * 16 + lh_o * 8;
#pragma unroll
      for (int g = 0; g < PG; ++g) {
        const int rr = row0 + g, Y = Y0 + g;
        const bool pv = colv && rr >= 1 && rr <= PT / 32 - 2 && Y >= 0 && Y < P.h;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
          typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
          u32x2 hq[2], lq[2];  // HF_H_SPLIT_STORE_PAIR: the split results of the pixels (2X, 2X+1), stored after the pc loop
          bool ovf_a = false;
#pragma unroll
          for (int pc = 0; pc < 2; ++pc) {
            // the H rows 2Y+a-1 .. 2Y+a+2 of this column: (Y-1, 1), (Y, 0), (Y, 1), (Y+1, 0), (Y+1, 1)
            float hm1[4], h2[4], h3[4];
            if (HF_H_EPI_ABLATE & 4) {
#pragma unroll
              for (int k = 0; k < 4; ++k) hm1[k] = H[pc][g][k], h2[k] = H[2 + pc][g][k], h3[k] = H[pc][g][k];
            } else {
            if (g == 0 && a == 0) {
              const float4 u = above[(ABOVE_P1 + pc) * 64];
              hm1[0] = u.x; hm1[1] = u.y; hm1[2] = u.z; hm1[3] = u.w;
            }
            if (g == PG - 1) {
              const float4 d0 = below[pc * 64];
              h2[0] = d0.x; h2[1] = d0.y; h2[2] = d0.z; h2[3] = d0.w;
              if (a == 1) {
                const float4 d1 = below[(2 + pc) * 64];
                h3[0] = d1.x; h3[1] = d1.y; h3[2] = d1.z; h3[3] = d1.w;
              }
            }
            }
            const float4 dm = *reinterpret_cast<const float4 *>(ep + c4), bs = *reinterpret_cast<const float4 *>(ep + CT + c4);
            const float dmv[4] = {dm.x, dm.y, dm.z, dm.w}, bsv[4] = {bs.x, bs.y, bs.z, bs.w};
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float h_m1 = (g == 0) ? hm1[k] : H[2 + pc][g == 0 ? 0 : g - 1][k];
              const float h_0 = H[pc][g][k], h_1 = H[2 + pc][g][k];
              const float h_2 = (g == PG - 1) ? h2[k] : H[pc][g == PG - 1 ? g : g + 1][k];
              const float h_3 = (g == PG - 1) ? h3[k] : H[2 + pc][g == PG - 1 ? g : g + 1][k];
              float o = (HF_H_EPI_ABLATE & 8) ? (a == 0 ? h_0 : h_1)
                        : (a == 0) ? fmaf(kyf[3], h_2, fmaf(kyf[2], h_1, fmaf(kyf[1], h_0, kyf[0] * h_m1)))
                                   : fmaf(kyf[3], h_3, fmaf(kyf[2], h_2, fmaf(kyf[1], h_1, kyf[0] * h_0)));
              o = fmaf(o, dmv[k], fmaf(nw_f, nz[g][a * 2 + pc], bsv[k]));  // d and bias carry the output scale (ep_fold)
              v[k] = fmaxf(o, o * P.alpha);
            }
            const int pix = pix00 + (2 * g + a) * (2 * P.w) + pc;
            if (!SPLIT) {
              if (pv) {
                float *ob = of_b + (long long)c4 * oplane + pix;
#pragma unroll
                for (int k = 0; k < 4; ++k) ob[k * oplane] = v[k];
              }
            } else {
              // every lane splits (the range vote inside hf_split4_f16 is wave-wide), valid positions store
              const float4 sn = *reinterpret_cast<const float4 *>(ep + 2 * CT + c4);
              const float vs[4] = {v[0] * sn.x, v[1] * sn.y, v[2] * sn.z, v[3] * sn.w};
              hf_half4 h4, l4;
              bool ovf = false;
              if (HF_H_EPI_ABLATE & 2) {
#pragma unroll
                for (int k = 0; k < 4; ++k) h4[k] = (_Float16)vs[k];
                l4 = h4;
              } else {
                hf_split4_f16(vs, h4, l4, ovf);
              }
#if HF_H_SPLIT_STORE16
              static_assert(NTERMS == 3, "HF_H_SPLIT_STORE16 pairs the hi and lo units of a pixel");
              // one 16-byte store per lane: the half-waves trade halves, lanes 0-31 write the hi unit of the pixel,
              // lanes 32-63 its lo unit (both halves of a wave share li, i.e. the pixel and its validity)
              typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
              typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
              u32x2 hu = __builtin_bit_cast(u32x2, h4), lu = __builtin_bit_cast(u32x2, l4);
              unsigned a0 = hu.x, b0 = lu.x, a1 = hu.y, b1 = lu.y;
              hf_half_swap(a0, b0);
              hf_half_swap(a1, b1);
              if (pv) {
                ovf_tile = ovf_tile || ovf;
                u32x4 unit;
                unit.x = a0; unit.y = a1; unit.z = b0; unit.w = b1;
                *reinterpret_cast<u32x4 *>((lh_o ? ol_b : oh_b) + (long long)q * oplane * 16 + (long long)pix * 16) = unit;
              }
#elif HF_H_SPLIT_STORE_PAIR
              hq[pc] = __builtin_bit_cast(u32x2, h4);
              lq[pc] = __builtin_bit_cast(u32x2, l4);
              ovf_a = ovf_a || ovf;
#else
              if (pv && !((HF_H_EPI_ABLATE & 1) && P.alpha != 77.0f)) {
                ovf_tile = ovf_tile || ovf;
                *reinterpret_cast<hf_half4 *>(oh_b + cb_ofs + (long long)pix * 16) = h4;
                if (NTERMS == 3) *reinterpret_cast<hf_half4 *>(ol_b + cb_ofs + (long long)pix * 16) = l4;  // plain fp16 consumer: no lo part
              }
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
          }
#if HF_H_SPLIT_STORE_PAIR && !HF_H_SPLIT_STORE16
          if (SPLIT) {
            // lanes i / i+32 hold channels 0-3 / 4-7 of the same two pixels: after the swap the low half-wave owns the whole unit
            // of pixel 2X, the high half-wave that of pixel 2X+1 - (a0, a1) = channels 0-3, (b0, b1) = channels 4-7 in both
            const long long unit_ofs = (long long)q * oplane * 16 + (long long)(pix00 + (2 * g + a) * (2 * P.w) + lh_o) * 16;
            unsigned a0 = hq[0].x, b0 = hq[1].x, a1 = hq[0].y, b1 = hq[1].y;
            hf_half_swap(a0, b0);
            hf_half_swap(a1, b1);
            u32x4 uh;
            uh.x = a0; uh.y = a1; uh.z = b0; uh.w = b1;
            u32x4 ul = uh;
            if (NTERMS == 3) {  // plain fp16 consumer: no lo part
              unsigned c0 = lq[0].x, d0 = lq[1].x, c1 = lq[0].y, d1 = lq[1].y;
              hf_half_swap(c0, d0);
              hf_half_swap(c1, d1);
              ul.x = c0; ul.y = c1; ul.z = d0; ul.w = d1;
            }
            if (pv && !((HF_H_EPI_ABLATE & 1) && P.alpha != 77.0f)) {
              ovf_tile = ovf_tile || ovf_a;
              *reinterpret_cast<u32x4 *>(oh_b + unit_ofs) = uh;
              if (NTERMS == 3) *reinterpret_cast<u32x4 *>(ol_b + unit_ofs) = ul;
            }
            __builtin_amdgcn_sched_barrier(0);
          }
#endif
        }
      }
      HF_TRACE_POINT(33);  // vertical pass + tail + stores of the quad issued
      if (!(HF_H_EPI_ABLATE & 4) || q == 3) hf_barrier_lds();  // everyone has read: the region may be overwritten (next quad / next tile's DMA)
      HF_TRACE_POINT(34);
    }
    hf_note_overflow(ovf_tile);
  };

  // (s_setprio 1 for the second-dispatched half of the block - MI355X_MICROARCH.md, "two waves per SIMD", item 4 -
  // measured no effect on any generator layer)
  // PP ("ping-pong"; pre-split input = all staging is LDS-DMA, eight waves = two per SIMD): the two waves of a SIMD (w and
  // w + 4: a workgroup's waves go round the four SIMDs) take TURNS on the matrix pipe.  Phase A of a K stage: waves 0-3 run the
  // stage's MFMAs back to back while waves 4-7 issue their LDS-DMA copies of the next stage; s_barrier; phase B: roles swapped;
  // s_barrier.  In the one-phase form every wave alternates a tap's side work with its MFMAs and all eight run in lock-step:
  // the pipe idles while both waves of a SIMD sit in their DMA issues (100-185 cycles each for an in-order wave; a tap-step took
  // 1.2-1.5 k cycles for 768 cycles of MFMA time, profiles/r04a_trace_fused.txt) - here the partner's MFMAs cover them
  // (MI355X_MICROARCH.md, "Two waves per SIMD": matrix beside memory pays, matrix beside matrix does not).  Per wave the K order
  // is unchanged: equal bits.  Hazards: stage c is complete since the end-of-stage barrier of stage c-1 (every wave drained its
  // copies, vmcnt(0)); the copies of stage c+1 go to the buffer stage c-1 was read from, whose last readers (phase B of c-1)
  // passed that barrier too; the mid-stage barrier only swaps the roles (copies stay in flight across it).
  const int pp_half = wave >> 2;
  int stage = 0;  // LDS buffer = stage & 1, running across tiles
  while (true) {
    HF_TRACE_POINT(1);  // tile start
    const TileGeom G = geom(cur.gi);
    const int tw = 1 << G.lg_tw, th = 1 << G.lg_th, wp = tw + HALO;
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
    float nzr[PG];
    float nzf[PG][4];  // FUSE: noise of output pixels (2Y+a, 2X+b), index a*2+b
#pragma unroll
    for (int g = 0; g < PG; ++g) {
      nzr[g] = 0.0f;
#pragma unroll
      for (int k = 0; k < 4; ++k) nzf[g][k] = 0.0f;
    }
    HF_TRACE_POINT(7);  // tile geometry done
    // the block's next tile, and its image's s into the other slot (read only after
    // the barriers of this tile's first nchunks-1 stages; nchunks >= 2)
    const int t_next = t_cur + grid_x;
    const bool has_next = t_next < P.n_tiles;
    Tile nxt = cur;
    int nxt_sl_off = sl_off;
    if (has_next) {
      nxt = locate(t_next);
      if (nxt.b0 != cur.b0) {
        if constexpr (LATE_TABLES) {
          sl_slot ^= 1;
          nxt_sl_off = sl_slot * P.cin;
          load_tables_issue(nxt.b0);
          pend = true;
        } else {
          // the slot about to be overwritten held the image before this one: another wave may still
          // be in that tile's epilogue (reading its d / bias) when images change on every tile
          hf_barrier_lds();
          sl_slot ^= 1;
          nxt_sl_off = sl_slot * P.cin;
          load_s(nxt.b0, sl_slot);
        }
      }
    }
    HF_TRACE_POINT(8);  // next tile located

    // one chunk = 9 tap-steps; side work spread over the steps: step 0 issues the activation
    // loads of the next stage into registers, steps 0-2 its weight DMAs, the last steps convert
    // the activations (6+ steps of MFMA time after their loads) into the other buffer
    for (int c = 0; c < nchunks; ++c, ++stage) {
      const int cb = stage & 1;
      half8 *buf = lds + cb * BUF_UNITS, *nbuf = lds + (cb ^ 1) * BUF_UNITS;
      const bool last = c + 1 == nchunks;
      const bool more1 = !last || has_next;
      const int cpf = last ? 0 : c + 1;  // chunk being staged (of this tile, or chunk 0 of the next)
      if (last && has_next) {  // switch the prefetch source to the next tile
        locate_items(nxt, e_src);
        xb = P.x + (long long)nxt.b0 * P.cin * plane;
        if (PRE) {
          xh_b = static_cast<const char *>(P.xh) + (long long)nxt.b0 * (P.cin / 8) * plane * 16;
          xl_b = static_cast<const char *>(P.xl) + (long long)nxt.b0 * (P.cin / 8) * plane * 16;
        }
        sl_off = nxt_sl_off;
      }
      if (last && !UP && P.noise) {  // this tile's noise, in registers before the epilogue (no loads there)
#pragma unroll
        for (int g = 0; g < PG; ++g) {
          const int p = (wave_pg + g) * 32 + li;
          const int Y = min(cur.ty0 + ((p >> G.lg_tw) & (th - 1)), P.h - 1), X = min(cur.tx0 + (p & (tw - 1)), P.w - 1);
          nzr[g] = P.noise[(long long)cur.b0 * P.noise_bstride + (long long)Y * P.out_w + X];
        }
      }
      if (FUSE && last && P.noise) {  // the 2x2 output pixels of the lane's position, per row group
#pragma unroll
        for (int g = 0; g < PG; ++g) {
          const int Y = min(max(cur.ty0 + wave_pg + g, 0), P.h - 1), X = min(max(cur.tx0 + li, 0), P.w - 1);
          const float *np = P.noise + (long long)cur.b0 * P.noise_bstride + (long long)(2 * Y) * (2 * P.w) + 2 * X;
          const float2 n0 = *reinterpret_cast<const float2 *>(np), n1 = *reinterpret_cast<const float2 *>(np + 2 * P.w);
          nzf[g][0] = n0.x; nzf[g][1] = n0.y; nzf[g][2] = n1.x; nzf[g][3] = n1.y;
        }
      }
      const half8 *a_hi = buf + lh * CT + wave_co + li;  // + tap*2*CT + ct*32
      const half8 *b_hi = buf + OFF_XH + lh * NPIX;      // + pixrow + tap column
      // fragments of tap+1 are fetched from LDS while the MFMAs of tap run
      // UP with in-kernel staging: 128 accumulator registers + the staging registers leave no room
      // for a second fragment set
#ifndef HF_H_DEEP_FETCH
#define HF_H_DEEP_FETCH 0  // experiment: same-resolution kernels fetch the LDS fragments TWO taps ahead (three slots)
#endif
      constexpr int NSLOT = (UP && !PRE) ? 1 : ((!UP && HF_H_DEEP_FETCH) ? 3 : 2);
      // PRE: all DMAs of the next stage in the first tap-step (short K loops: HBM latency exceeds the stage's MFMA
      // time, the copies need the whole stage to land) or spread one per tap-step (long K loops, see DMA_PER_STEP)
      const bool early = PRE && P.dma_early;
      half8 ah[NSLOT][CT_TILES], al[NSLOT][CT_TILES], bh[NSLOT][PG], bl[NSLOT][PG];
      // part: 0 = hi and lo (program order hi0 lo0 hi1 lo1), 1 = hi only, 2 = lo only (HF_H_FETCH_ORDER: the reads in the order the
      // MFMAs consume them - a-hi, b-hi, b-lo, a-lo)
      auto fetch_a = [&](int slot, int tap, int part = 0) {
#pragma unroll
        for (int ct = 0; ct < CT_TILES; ++ct) {
          if (part != 2) ah[slot][ct] = a_hi[tap * 2 * CT + ct * 32];
          if (NTERMS == 3 && part != 1) al[slot][ct] = a_hi[OFF_WL + tap * 2 * CT + ct * 32];
        }
      };
      auto fetch_b = [&](int slot, int tap, int part = 0) {
        const int ky = tap / 3, kx = tap % 3;
        const int brow = UP ? (ky == 2 ? 0 : 1) : ky, bcol = UP ? (kx == 2 ? 0 : 1) : kx;
#pragma unroll
        for (int g = 0; g < PG; ++g) {
          if (part != 2) bh[slot][g] = b_hi[pixrow[g][brow] + bcol];
          if (NTERMS == 3 && part != 1) bl[slot][g] = b_hi[X_UNITS + pixrow[g][brow] + bcol];
        }
      };
      if (PP && pp_half == 1) {  // phase A of the second half: its copies of the next stage, then the role swap
        if (more1) {
#pragma unroll
          for (int j = 0; j < ND; ++j) dma_piece(j, cpf, cb ^ 1);
#pragma unroll
          for (int e = 0; e < XE; ++e) dma_x(e, cpf, cb ^ 1);
        }
        if (HF_H_PP_PREFETCH) {  // the stage has been complete since the last end-of-stage barrier: the first tap's fragments
                                 // travel while the other half still computes, not after the role swap
          fetch_a(0, tap_at<UP>(0));
          fetch_b(0, tap_at<UP>(0));
        }
        hf_barrier_lds();
      }
      if (!(PP && pp_half == 1 && HF_H_PP_PREFETCH)) {
        fetch_a(0, tap_at<UP>(0));
        fetch_b(0, tap_at<UP>(0));
      }
      if (PP && HF_H_PP_EARLY_X > 0 && pp_half == 0 && more1) {  // first half: some of its activation copies under the latency of its first fragments
#pragma unroll
        for (int e = 0; e < XE; ++e)
          if (e < HF_H_PP_EARLY_X) dma_x(e, cpf, cb ^ 1);
      }
      if (NSLOT == 3) {
        fetch_a(1, 1);
        fetch_b(1, 1);
      }
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const int tap = tap_at<UP>(i);  // the side work below is scheduled by POSITION i
        HF_TRACE_POINT(10 + i);
        const int grp = tap_group<UP>(i);
        const int sa = (NSLOT == 1) ? 0 : (NSLOT == 3 ? i % 3 : (i & 1)), sb = (NSLOT == 1) ? 0 : (NSLOT == 3 ? i % 3 : (grp & 1));
        if (NSLOT == 1) {
          if (i > 0) fetch_a(0, tap);
          if (i > 0 && group_first<UP>(grp) == i) fetch_b(0, tap);
        } else if (NSLOT == 3) {  // !UP: natural tap order, one fragment pair per tap
          if (i + 2 < 9) {
            if (HF_H_FETCH_ORDER) {
              fetch_a((i + 2) % 3, i + 2, 1);
              fetch_b((i + 2) % 3, i + 2, 1);
              fetch_b((i + 2) % 3, i + 2, 2);
              fetch_a((i + 2) % 3, i + 2, 2);
            } else {
              fetch_a((i + 2) % 3, i + 2);
              fetch_b((i + 2) % 3, i + 2);
            }
          }
        } else {
          const bool nb = group_first<UP>(grp) == i && group_first<UP>(grp + 1) < 9;
          if (HF_H_FETCH_ORDER) {
            if (i + 1 < 9) fetch_a(sa ^ 1, tap_at<UP>(i + 1), 1);
            if (nb) fetch_b(sb ^ 1, tap_at<UP>(group_first<UP>(grp + 1)), 1);
            if (nb) fetch_b(sb ^ 1, tap_at<UP>(group_first<UP>(grp + 1)), 2);
            if (i + 1 < 9) fetch_a(sa ^ 1, tap_at<UP>(i + 1), 2);
          } else {
            if (i + 1 < 9) fetch_a(sa ^ 1, tap_at<UP>(i + 1));
            // the next group's activation fragment, as soon as its slot is free (= when this group starts)
            if (nb) fetch_b(sb ^ 1, tap_at<UP>(group_first<UP>(grp + 1)));
          }
        }
        // ---- side work of this step (PP: none - the copies are issued in the wave's other phase) ----
        if (!PP && more1) {
#pragma unroll
          for (int j = 0; j < ND; ++j)
            if ((early ? 0 : j / DMA_PER_STEP) == i) dma_piece(j, cpf, cb ^ 1);
        }
        if (!PP && PRE && more1) {  // activation DMAs in tap-steps 1, 3, 5, ...
#pragma unroll
          for (int e = 0; e < XE; ++e)
            if (i == (early ? 0 : ((2 * XE <= 9) ? 1 + 2 * e : 1 + e))) dma_x(e, cpf, cb ^ 1);
        }
        if (!PRE && more1 && i == 0) {
#pragma unroll
          for (int e = 0; e < XE; ++e) load_item(e, cpf);
        }
        // (staggering the conversions over the two waves of a SIMD - early half in taps 3-5 - was
        // measured 5-15 % slower: the early half then waits on its loads)
        if (more1 && i == 9 - XE) HF_TRACE_POINT(5);  // before the first conversion (waits for the loads)
        if (!PRE && more1 && i >= 9 - XE) convert_item(i - (9 - XE), cpf, nbuf);
        if (more1 && i == 8) HF_TRACE_POINT(6);  // conversions done
        constexpr bool ILV = HF_H_ILV && PP && NSLOT >= 2;
        if (!ILV) __builtin_amdgcn_sched_barrier(0);
        HF_TRACE_POINT(20 + i);  // side work issued, before the MFMAs
        if (PP && HF_H_PP_EARLY_BAR && i == 8) {
          // the half's LAST tap: its fragments are in registers (the barriers wait for them), nothing of this stage is read from
          // LDS any more - the barrier the other half waits at is passed BEFORE the tap's MFMAs instead of behind them: the other
          // half's start-up (role swap: its first MFMAs; end of stage: the request of its first fragments) runs under them
          if (pp_half == 0) hf_barrier_lds();
          else HF_H_BARRIER();
        }
        const int ph = UP ? (((tap / 3) & 1) * 2 + ((tap % 3) & 1)) : 0;
#pragma unroll
        for (int ct = 0; ct < CT_TILES; ++ct)
#pragma unroll
          for (int g = 0; g < PG; ++g)
            acc[ph][ct][g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[sa][ct], bh[sb][g], acc[ph][ct][g], 0, 0, 0);
        if (NTERMS == 3) {
#pragma unroll
          for (int ct = 0; ct < CT_TILES; ++ct)
#pragma unroll
            for (int g = 0; g < PG; ++g)
              acc[ph][ct][g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[sa][ct], bl[sb][g], acc[ph][ct][g], 0, 0, 0);
#pragma unroll
          for (int ct = 0; ct < CT_TILES; ++ct)
#pragma unroll
            for (int g = 0; g < PG; ++g)
              acc[ph][ct][g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[sa][ct], bh[sb][g], acc[ph][ct][g], 0, 0, 0);
        }
        if constexpr (ILV) {
          constexpr int NRA = CT_TILES * (NTERMS == 3 ? 2 : 1), NRB = PG * (NTERMS == 3 ? 2 : 1), NM = CT_TILES * PG * NTERMS;
          const bool fb = group_first<UP>(grp) == i && group_first<UP>(grp + 1) < 9;  // this step fetched the next group's B fragments
          if (NSLOT == 3) {
            if (i + 2 < 9) hf_interleave<(NRA + NRB < NM ? NRA + NRB : NM), NM>();
          } else if (i + 1 < 9 && fb) hf_interleave<(NRA + NRB < NM ? NRA + NRB : NM), NM>();
          else if (i + 1 < 9) hf_interleave<(NRA < NM ? NRA : NM), NM>();
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      HF_TRACE_POINT(2);  // chunk MFMAs issued, before the barrier
      if (PP && pp_half == 0) {  // phase B of the first half: role swap, then its copies of the next stage
        if (!HF_H_PP_EARLY_BAR) hf_barrier_lds();
        if (more1) {
#pragma unroll
          for (int j = 0; j < ND; ++j) dma_piece(j, cpf, cb ^ 1);
#pragma unroll
          for (int e = 0; e < XE; ++e)
            if (e >= HF_H_PP_EARLY_X) dma_x(e, cpf, cb ^ 1);
        }
      }
      // next stage complete (DMA landed, conversions written), current one free
      if (!(PP && HF_H_PP_EARLY_BAR && pp_half == 1)) HF_H_BARRIER();
      HF_TRACE_POINT(3);  // after the barrier
    }
    if (LATE_TABLES && pend) {  // the next image's tables: loaded at the head of this tile, they have long arrived
      load_tables_commit(sl_slot);
      pend = false;
    }

    if (FUSE) {
      if (P.oh) epilogue_fused(std::true_type{}, cur, ep_slot, nzf, ((stage - 1) & 1));
      else epilogue_fused(std::false_type{}, cur, ep_slot, nzf, ((stage - 1) & 1));
    } else {
      bool done = false;
      if constexpr (!UP) {
        if (fast_ep) {
          epilogue_fast(G, cur, ep_slot, nzr);
          done = true;
        }
      }
      if (!done) epilogue(G, cur, ep_slot, nzr);
    }
    HF_TRACE_POINT(4);  // epilogue issued
    if (!has_next) break;
    zero_acc();
    if (nxt.b0 != cur.b0) ep_slot ^= 1;
    cur = nxt;
    t_cur = t_next;
  }
}

// fp32 prepared weights wt[tap][ci][co] -> hi / lo halves in [chunk16][tap][kg][co][8], PRE-SCALED
// by a power of two 2^k chosen so that max|wt| * 2^k lies in [2^13, 2^14): prepared weights are
// scale*W ~ 1e-2, whose lo parts would otherwise be fp16 subnormals (absolute error 2^-25 instead
// of a relative 2^-22, see hf_split_f16).  With the pre-scale every weight within 2^-15 of the
// largest keeps its full 22 bits.  The inverse 2^-k is stored in the 16-byte TRAILER that follows
// the 9*cin*cout halves of wt_hi (float trailer[0]; trailer[1] = bit pattern of max|wt|) and is
// folded into the epilogue's output scale by the conv kernels (exact: a power of two).
__global__ __launch_bounds__(256) void split_weights_absmax(unsigned int *__restrict__ trailer,
                                                            const float *__restrict__ wt, long long n) {
  float m = 0.0f;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) m = fmaxf(m, fabsf(wt[i]));
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
  // one atomic per wave, and only when it can still raise the maximum (a plain read first: thousands of
  // waves hammering one address with atomics took 190 us per call)
  if ((threadIdx.x & 63) == 0 && __float_as_uint(m) > trailer[1]) atomicMax(trailer + 1, __float_as_uint(m));
}
__global__ void split_weights_clear(unsigned int *trailer) {
  if (threadIdx.x < 4) trailer[threadIdx.x] = 0u;
}
__device__ __forceinline__ int weight_prescale_log2(unsigned int absmax_bits) {
  const int e = (int)((absmax_bits >> 23) & 255u) - 127;  // floor(log2 max|wt|)
  if (e <= -127 || e >= 128) return 0;                    // all zero / denormal / inf: leave alone
  int k = 13 - e;
  return k > 120 ? 120 : (k < -120 ? -120 : k);
}
__global__ __launch_bounds__(256) void split_weights(_Float16 *__restrict__ wth, _Float16 *__restrict__ wtl,
                                                     const float *__restrict__ wt, int cin, int cout, int taps) {
  const long long n = (long long)taps * cin * cout;
  unsigned int *trailer = reinterpret_cast<unsigned int *>(wth + n);
  const int k = weight_prescale_log2(trailer[1]);
  const float up = __uint_as_float((unsigned int)(127 + k) << 23);
  if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<float *>(trailer)[0] = __uint_as_float((unsigned int)(127 - k) << 23);
  const long long stride = (long long)gridDim.x * blockDim.x;
  bool ovf = false;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int co = (int)(i % cout);
    const long long r = i / cout;
    const int ci = (int)(r % cin), tap = (int)(r / cin);
    const long long dst = ((((long long)(ci / 16) * taps + tap) * 2 + (ci % 16) / 8) * cout + co) * 8 + (ci % 8);
    _Float16 h, l;
    hf_split_f16(wt[i] * up, h, l, ovf);
    wth[dst] = h;
    if (wtl) wtl[dst] = l;
  }
}

template <int NTERMS, int CT_TILES, int PG, int WAVES_CO, int WAVES_PX, bool UP, int TWMAX = 32, bool PRE = false, bool FUSE = false>
int launch_h(ConvParams &P, const _Float16 *wth, const _Float16 *wtl, hipStream_t st) {
  constexpr int NT = 64 * WAVES_CO * WAVES_PX;
  constexpr int CT = 32 * CT_TILES * WAVES_CO;
  constexpr int PT = 32 * PG * WAVES_PX;
  constexpr int NPIX = halo_pixels_max<PT, UP, TWMAX, FUSE>();
  constexpr int NPART = (NTERMS == 3) ? 2 : 1;
  if (P.cin % KH || P.cout % CT || P.stride != 1 || P.t || P.groups > 1 || P.residual || P.act == ACT_PRELU)
    return HF_E_INVALID;
  if ((long long)P.cin * P.h * P.w >= (1LL << 31)) return HF_E_INVALID;
  if ((long long)CT * P.out_h * P.out_w * 4 >= (1LL << 32)) return HF_E_INVALID;  // 32-bit epilogue offsets
  P.splits = 1;
  P.n_geom = 1;
  P.g[0] = make_geom(0, 0, P.h, P.w, P.batch, PT, 0);
  if (TWMAX > 32 && P.w >= 64 && P.g[0].lg_nb == 0) {  // wider tiles: longer contiguous row segments
    const int tw = min(TWMAX, pow2_floor(P.w)), th = PT / tw;
    if (th >= 1 && P.h >= th) {
      P.g[0].lg_tw = ilog2(tw); P.g[0].lg_th = ilog2(th);
      P.g[0].tiles_x = hf_cdiv(P.w, tw); P.g[0].tiles_y = hf_cdiv(P.h, th);
    }
  }
  if (FUSE) {  // overlapping th x 32 tiles, origin -1, (th-2) x 30 interior positions each (see the kernel)
    constexpr int TH = PT / 32;
    P.g[0].lg_tw = 5; P.g[0].lg_th = ilog2(TH); P.g[0].lg_nb = 0;
    P.g[0].tiles_x = hf_cdiv(P.w, 30); P.g[0].tiles_y = hf_cdiv(P.h, TH - 2); P.g[0].tiles_b = P.batch;
  }
  int nblocks = geom_blocks(P.g[0]);
  if (UP && !FUSE) {  // + the Y = h row (incl. corner) and the X = w column of the (h+1)x(w+1) phase domain
    P.n_geom = 3;
    constexpr int RIM = PT >= 512 ? 4 : 2;
    P.g[1] = make_geom(P.h, 0, 1, P.w + 1, P.batch, PT, nblocks, true, RIM);
    nblocks += geom_blocks(P.g[1]);
    P.g[2] = make_geom(0, P.w, P.h, 1, P.batch, PT, nblocks, true, RIM);
    nblocks += geom_blocks(P.g[2]);
  }
  for (int i = 0; i < P.n_geom; ++i) {
    // one image per tile, whole PT-pixel tiles, halo within the LDS tile
    if (P.g[i].lg_nb != 0 || (1 << (P.g[i].lg_tw + P.g[i].lg_th)) != PT) return HF_E_INVALID;
    if (geom_xs(P.g[i], 1, UP ? 1 : 2) > NPIX) return HF_E_INVALID;
  }
  constexpr int NW_ = WAVES_CO * WAVES_PX;
  constexpr int XSLOTS = (PG == 1) ? 4 : 6;
  constexpr int BUF_UNITS = (FUSE && NPART * (9 * 2 * CT + 2 * NPIX) < NW_ * XSLOTS * 64) ? NW_ * XSLOTS * 64 : NPART * (9 * 2 * CT + 2 * NPIX);
  const size_t lds = (size_t)2 * BUF_UNITS * 16 + 2 * ((P.cin + 3) & ~3) * sizeof(float) +
                     2 * 3 * CT * sizeof(float) + (P.rgb_out ? 2 * 3 * CT * sizeof(float) : 0);
  // stages + s[2][cin] + epilogue d/bias/s_next [2][3][CT] (+ fused ToRGB weights [2][3][CT])
  // fused ToRGB: the standard StyledConv tail (the kernel's fast epilogue) writes one raw slab per 32*CT_TILES output
  // channels; the general epilogue only handles the one-slab case
  const bool std_tail = P.bias && P.act == ACT_LRELU && P.alpha >= 0.0f && P.alpha <= 1.0f && P.scale > 0.0f;
  if (P.rgb_out && (UP || !P.rgb_w || !P.rgb_s || (!std_tail && (WAVES_CO != 1 || P.cout != CT)) ||
                    P.rgb_slabs != P.cout / (32 * CT_TILES)))
    return HF_E_INVALID;
  if (lds > 160 * 1024) return HF_E_INVALID;
  P.n_tiles = nblocks;
  P.dma_early = (g_h_tune & 1) ? 1 : 0;  // measured (tools/probes/gen_layers.py): spread is 0-8 % faster on every generator layer
  // LDS allows one block per CU: size the grid to the chip and let each block walk its share
  // of the tiles as one pipeline (the tile-to-tile hand-over needs >= 2 stages per tile)
  const int co_tiles = P.cout / CT;
  const int per_cu = (2 * lds <= 160 * 1024) ? 2 : 1;
  int resident = (g_h_blocks > 0 ? g_h_blocks : 256 * per_cu) / co_tiles;
  if (resident < 1) resident = 1;
  // tiles with many K stages (cin > 128) gain nothing from the cross-tile hand-over and measured
  // 0-13 % faster as one block per tile (512->512 @ 64^2: 372 -> 428 TF/s); the short ones
  // (cin <= 128: 2-8 stages per tile) are 1.5-4 % faster walked by resident blocks
  const bool walk = P.cin / KH >= 2 && (P.cin / KH <= 8 || g_h_blocks > 0);
  const int gx = (walk && nblocks > resident) ? resident : nblocks;
  dim3 grid(gx, co_tiles);
  if (grid.y > 65535) return HF_E_INVALID;
  // Block order (see csrc/convh_enc.hip launch_enc): tiles-fastest re-reads every input tile once per cout tile from beyond L2,
  // cout-tiles-fastest streams the weights once per group of resident input tiles instead - taken when that moves fewer bytes
  P.swap_xy = 0;
  if (HF_H_SWAP_XY && co_tiles > 1 && gx <= 65535) {
    const double in_bytes = (double)P.batch * P.cin * P.h * P.w * 4.0, w_bytes = 9.0 * P.cin * P.cout * 4.0;
    const double tiles_fast = co_tiles * in_bytes + w_bytes;
    const double resident_tiles = co_tiles >= 256 ? 1.0 : 256.0 / co_tiles;
    const double cols_fast = in_bytes + w_bytes * ((double)nblocks / resident_tiles);
    // (inputs that fit the 256 MB Infinity Cache beside everything else are re-read from there either way: measured neutral at
    // batch 8 - tools/probes/gen_layers.py, r05j - so only from half of it upward, i.e. in the batched swap's generator calls)
    if ((g_h_tune & 8) || (in_bytes > 128e6 && cols_fast < 0.5 * tiles_fast)) {
      P.swap_xy = 1;
      grid = dim3(co_tiles, gx);
    }
  }
  if (PRE) {
    if (!P.xh || (NTERMS == 3 && !P.xl) || P.s || (P.cin & 15)) return HF_E_INVALID;
    if ((long long)2 * P.h * P.w * 16 >= (1LL << 31)) return HF_E_INVALID;  // 32-bit offsets inside a stage
    hipLaunchKernelGGL((conv_mfma_h<NTERMS, CT_TILES, PG, WAVES_CO, WAVES_PX, false, UP, TWMAX, PRE, FUSE>), grid, dim3(NT), lds, st,
                       P, wth, wtl);
  } else if (P.s) {
    hipLaunchKernelGGL((conv_mfma_h<NTERMS, CT_TILES, PG, WAVES_CO, WAVES_PX, true, UP, TWMAX, false, FUSE>), grid, dim3(NT), lds, st,
                       P, wth, wtl);
  } else {
    if (FUSE) return HF_E_INVALID;  // the generator always modulates
    hipLaunchKernelGGL((conv_mfma_h<NTERMS, CT_TILES, PG, WAVES_CO, WAVES_PX, false, UP, TWMAX, false, false>), grid, dim3(NT), lds,
                       st, P, wth, wtl);
  }
  return hf_launch_status();
}

}  // namespace

namespace hf_detail {

// tuning / test hooks: per-thread state (no process-global mutable state in the library)
thread_local int g_force_h = 0;
thread_local int g_h_blocks = 0;
thread_local int g_h_tune = 0;
int g_batch_invariant = 0;  // process-wide: a mode of the whole library, not a per-thread debug hook

int launch_conv_h(ConvParams &P, int nterms, bool up, const void *wth, const void *wtl, hipStream_t st) {
  const _Float16 *h = static_cast<const _Float16 *>(wth), *l = static_cast<const _Float16 *>(wtl);
  if (!h || (nterms == 3 && !l)) return HF_E_INVALID;
  int cfg, rc;
  if (up) {
    // 61: 64 co x 256 positions x 4 phases, 2 co-waves x 4 pixel-waves, 1x2 MFMA tiles per phase
    // 63: 32 co x 512 positions x 4 phases, 8 pixel-waves (cout % 64 != 0: the 1024^2 layer)
    cfg = (P.cout % 64) ? 63 : 61;
    if (P.xh) {  // pre-split activations (ids 8x)
      if (cfg == 63) rc = (nterms == 3) ? launch_h<3, 1, 2, 1, 8, true, 32, true>(P, h, l, st) : launch_h<1, 1, 2, 1, 8, true, 32, true>(P, h, l, st);
      else rc = (nterms == 3) ? launch_h<3, 1, 2, 2, 4, true, 32, true>(P, h, l, st) : launch_h<1, 1, 2, 2, 4, true, 32, true>(P, h, l, st);
      if (rc == HF_OK) note_path(5, cfg + 20);
      return rc;
    }
    if (cfg == 63) rc = (nterms == 3) ? launch_h<3, 1, 2, 1, 8, true>(P, h, l, st) : launch_h<1, 1, 2, 1, 8, true>(P, h, l, st);
    else rc = (nterms == 3) ? launch_h<3, 1, 2, 2, 4, true>(P, h, l, st) : launch_h<1, 1, 2, 2, 4, true>(P, h, l, st);
    if (rc == HF_OK) note_path(5, cfg);
    return rc;
  }
  // 51: 64 co x 256 px (8 rows), 2 co-waves x 4 pixel-waves, 1x2 MFMA tiles per wave
  // 52: 64 co x 512 px (16 rows), 8 pixel-waves, 2x2 MFMA tiles per wave: 0.67 LDS fragment
  //     reads per MFMA instead of 2 (fewer issue slots beside the MFMAs), needs >= 256 such blocks
  // 53: 32 co x 512 px (16 rows), 8 pixel-waves, 1x2 tiles: layers with cout % 64 != 0 (1024^2: 32)
  // 55/56: 53 with tiles of 128 / 64-pixel rows (4 / 8 rows): longer contiguous row segments for the
  //     HBM-bound 1024^2 layer (+9 % measured)
  // 54: 32 co x 256 px, 4 pixel-waves (256 threads), 80 KB of LDS: TWO resident blocks per CU, so
  //     one block's barrier drains (activation loads, epilogue stores) hide under the other's
  //     MFMAs - the HBM-bound high-resolution layers (few chunks per tile)
  cfg = g_force_h;
  if (cfg == 0) {
    // (a tile FORM: same K order as 51, equal bits - tests/test_sim_kernels.py - so it follows the real launch in every mode)
    const long long blocks52 = (long long)P.batch * hf_cdiv(P.h, 16) * hf_cdiv(P.w, 32) * (P.cout / 64);
    if (P.cout % 64) cfg = (P.w >= 128) ? 55 : 53;
    else cfg = (P.h * P.w >= 512 && blocks52 >= 256) ? 52 : 51;
  }
  if (P.rgb_out && cfg == 51) cfg = 52;  // fused ToRGB needs all cout channels in one wave
  if (P.xh) {  // pre-split activations (ids 7x = the 5x tile shapes with DMA-staged activations)
    // 32 -> 32 channels (the 1024^2 layer): the row pipeline of convrow.hip (id 79); hf_debug_set_tuning bit 4 = the tiled form
    if (!g_force_h && !(g_h_tune & 16) && P.cin == 32 && P.cout == 32) {
      rc = launch_conv_rows(P, nterms, wth, wtl, st);
      if (rc == HF_OK) {
        note_path(5, 79);
        return rc;
      }
    }
    if (P.rgb_skip) return HF_E_INVALID;  // the finished ToRGB exists in the row pipeline only
    if (cfg == 55) rc = (nterms == 3) ? launch_h<3, 1, 2, 1, 8, false, 128, true>(P, h, l, st) : launch_h<1, 1, 2, 1, 8, false, 128, true>(P, h, l, st);
    else if (cfg == 53) rc = (nterms == 3) ? launch_h<3, 1, 2, 1, 8, false, 32, true>(P, h, l, st) : launch_h<1, 1, 2, 1, 8, false, 32, true>(P, h, l, st);
    else if (cfg == 52) rc = (nterms == 3) ? launch_h<3, 2, 2, 1, 8, false, 32, true>(P, h, l, st) : launch_h<1, 2, 2, 1, 8, false, 32, true>(P, h, l, st);
    else if (cfg == 51) rc = (nterms == 3) ? launch_h<3, 1, 2, 2, 4, false, 32, true>(P, h, l, st) : launch_h<1, 1, 2, 2, 4, false, 32, true>(P, h, l, st);
    else return HF_E_INVALID;
    if (rc == HF_OK) note_path(5, cfg + 20);
    return rc;
  }
  if (cfg == 55) rc = (nterms == 3) ? launch_h<3, 1, 2, 1, 8, false, 128>(P, h, l, st) : launch_h<1, 1, 2, 1, 8, false, 128>(P, h, l, st);
  else if (cfg == 56) rc = (nterms == 3) ? launch_h<3, 1, 2, 1, 8, false, 64>(P, h, l, st) : launch_h<1, 1, 2, 1, 8, false, 64>(P, h, l, st);
  else if (cfg == 54) rc = (nterms == 3) ? launch_h<3, 1, 2, 1, 4, false>(P, h, l, st) : launch_h<1, 1, 2, 1, 4, false>(P, h, l, st);
  else if (cfg == 53) rc = (nterms == 3) ? launch_h<3, 1, 2, 1, 8, false>(P, h, l, st) : launch_h<1, 1, 2, 1, 8, false>(P, h, l, st);
  else if (cfg == 52) rc = (nterms == 3) ? launch_h<3, 2, 2, 1, 8, false>(P, h, l, st) : launch_h<1, 2, 2, 1, 8, false>(P, h, l, st);
  else rc = (nterms == 3) ? launch_h<3, 1, 2, 2, 4, false>(P, h, l, st) : launch_h<1, 1, 2, 2, 4, false>(P, h, l, st);
  if (rc == HF_OK) note_path(5, cfg);
  return rc;
}

}  // namespace hf_detail

extern "C" int hf_conv_split_weights_f16(void *wt_hi, void *wt_lo, const float *wt, int cin, int cout, void *stream) {
  return hf_conv_split_weights_f16_taps(wt_hi, wt_lo, wt, cin, cout, 9, stream);
}

extern "C" int hf_conv_split_weights_f16_taps(void *wt_hi, void *wt_lo, const float *wt, int cin, int cout, int taps, void *stream) {
  if (!wt_hi || !wt || cin <= 0 || cout <= 0 || (cin % 16) || (taps != 9 && taps != 1)) return HF_E_INVALID;
  long long n = (long long)taps * cin * cout;
  long long g = (n + 255) / 256;
  if (g > 4096) g = 4096;
  unsigned int *trailer = reinterpret_cast<unsigned int *>(static_cast<_Float16 *>(wt_hi) + n);
  hipLaunchKernelGGL(split_weights_clear, dim3(1), dim3(64), 0, (hipStream_t)stream, trailer);
  hipLaunchKernelGGL(split_weights_absmax, dim3((int)(g > 512 ? 512 : g)), dim3(256), 0, (hipStream_t)stream, trailer, wt, n);
  hipLaunchKernelGGL(split_weights, dim3((int)g), dim3(256), 0, (hipStream_t)stream, static_cast<_Float16 *>(wt_hi),
                     static_cast<_Float16 *>(wt_lo), wt, cin, cout, taps);
  return hf_launch_status();
}

extern "C" unsigned long long hf_f16_overflow_count_convh(int reset) { return hf_f16_overflow_read_tu(reset); }

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

extern "C" int hf_modconv3x3_up_blur_f16_f32(float *out, void *split_hi, void *split_lo, const float *x, const void *x_hi,
                                             const void *x_lo, const void *wt_hi, const void *wt_lo, const float *s,
                                             const float *d, const float *blur_k1d_x, const float *blur_k1d_y, const float *noise, const float *noise_w,
                                             long long noise_bstride, const float *bias, const float *s_next, int batch, int cin,
                                             int cout, int h, int w, float alpha, float scale, void *stream) {
  // wt_lo NULL: plain fp16 operands (nterms 1, BASELINE.json configs[4]) - then x_lo / split_lo are not used either
  const int nterms = wt_lo ? 3 : 1;
  if ((!out == !split_hi) || (!x && !x_hi) || !wt_hi || !blur_k1d_x || !blur_k1d_y || batch <= 0 || cin <= 0 || cout <= 0 ||
      h < 2 || w < 2 || (noise && !noise_w) || (nterms == 3 && x_hi && !x_lo) || (!x_hi && !s) || (cout % 32) || (cin % 16) ||
      (split_hi && ((nterms == 3 && !split_lo) || (cout & 7))) || !bias || !(alpha >= 0.0f && alpha <= 1.0f) || !(scale > 0.0f))
    return HF_E_INVALID;  // exactly one output form; the epilogue assumes bias + leaky ReLU (0 <= alpha <= 1), scale > 0
  ConvParams P{};
  P.out = out; P.x = x; P.xh = x_hi; P.xl = x_lo; P.s = x_hi ? nullptr : s; P.d = d; P.noise = noise; P.noise_w = noise_w;
  P.bias = bias;
  P.s_bstride = cin; P.d_bstride = cout;
  P.groups = 1;
  P.noise_bstride = noise_bstride;
  P.batch = batch; P.cin = cin; P.cout = cout; P.h = h; P.w = w;
  P.out_h = 2 * h; P.out_w = 2 * w; P.out_wv = 2 * w;
  P.stride = 1;
  P.act = bias ? ACT_LRELU : ACT_NONE;
  P.alpha = alpha; P.scale = scale;
  P.oh = split_hi; P.ol = nterms == 3 ? split_lo : nullptr; P.s_next = s_next;
  for (int j = 0; j < 4; ++j) {  // upfirdn2d is a true convolution: flipped taps (op/upfirdn2d.py:186)
    P.blur_kx[j] = blur_k1d_x[3 - j];
    P.blur_ky[j] = blur_k1d_y[3 - j];
  }
  const _Float16 *hi = static_cast<const _Float16 *>(wt_hi), *lo = static_cast<const _Float16 *>(wt_lo);
  // <1,2,1,8>: 32 co x 512 positions (16 rows x 32), 8 waves x 2 rows, one block per CU - all eight waves in lock-step, so
  // the VALU-bound epilogue (29 k of a tile's 70 k cycles) and the MFMA loop never overlap.
  // (Round 4 measured <1,2,1,4> - 8 rows x 32, FOUR waves, TWO independent blocks per CU, 2 x 37.4 KB of stage buffers each
  // with the FUSE-aware halo size, so that a SIMD issues one block's epilogue under the other block's MFMAs - with both
  // blocks resident (round 2's "2x slower" was one block per CU: the rim-tile halo size made its LDS 86 KB): bit-identical,
  // 752 vs 602 us on the 1024^2 layer, 546 vs 432, 449 vs 424: every block stages the full weight stage - twice the LDS-DMA
  // issues per wave - and recomputes 43 % instead of 22 % halo.  Not kept.)
  int rc;
  // (Round 6, measured and rejected: <1,1,1,16> - the same tile as SIXTEEN waves x one row, four waves per SIMD at 128 registers,
  // so that the VALU-issue-bound epilogue runs at 2 instead of 3 cycles per instruction (tools/probes/valu_rate.hip): 806 / 519 /
  // 500 us against 670 / 440 / 400 - the A fragment is then read once per 3 instead of 6 MFMAs, the one-phase K loop returns, and
  // the kernel spills 212 bytes per lane.  profiles/r06g_fuse_16_waves.txt.  The kernel template still instantiates for it.)
  if (nterms == 3)
    rc = x_hi ? launch_h<3, 1, 2, 1, 8, true, 32, true, true>(P, hi, lo, (hipStream_t)stream)
              : launch_h<3, 1, 2, 1, 8, true, 32, false, true>(P, hi, lo, (hipStream_t)stream);
  else
    rc = x_hi ? launch_h<1, 1, 2, 1, 8, true, 32, true, true>(P, hi, lo, (hipStream_t)stream)
              : launch_h<1, 1, 2, 1, 8, true, 32, false, true>(P, hi, lo, (hipStream_t)stream);
  const int form = x_hi ? 93 : 73;
  if (rc == HF_OK) note_path(5, form);
  return rc;
}

extern "C" int hf_set_batch_invariant(int on) {
  const int prev = hf_detail::g_batch_invariant;
  hf_detail::g_batch_invariant = on ? 1 : 0;
  return prev;
}

extern "C" int hf_debug_set_tuning(int bits) {
  hf_detail::g_h_tune = bits;
  return HF_OK;
}

extern "C" int hf_modconv3x3_f16_rgb_slabs(int cout) {
  // the dispatch below gives a layer with a fused ToRGB the 64-channel-per-wave tile shape when cout % 64 == 0 (cfg 52),
  // else the 32-channel one (53 / 55)
  return cout <= 0 || (cout % 32) ? 0 : ((cout % 64) ? cout / 32 : cout / 64);
}

extern "C" int hf_debug_set_persistent_blocks(int blocks) {
  hf_detail::g_h_blocks = blocks;
  return HF_OK;
}

extern "C" int hf_modconv3x3_f16_rgb_f32(float *out, const float *x, const void *wt_hi, const void *wt_lo, int nterms,
                                         const float *s, const float *d, const float *noise, const float *noise_w,
                                         long long noise_bstride, const float *bias, int batch, int cin, int cout,
                                         int h, int w, float alpha, float scale, float *rgb_raw, const float *rgb_wt,
                                         const float *rgb_s, void *stream) {
  if (!out || !x || !wt_hi || batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0 || (noise && !noise_w) ||
      (nterms != 1 && nterms != 3) || !rgb_raw || !rgb_wt || !rgb_s || (cout % 32))
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
  P.rgb_out = rgb_raw; P.rgb_w = rgb_wt; P.rgb_s = rgb_s;
  P.rgb_slabs = hf_modconv3x3_f16_rgb_slabs(cout);
  return launch_conv_h(P, nterms, false, wt_hi, wt_lo, (hipStream_t)stream);
}

extern "C" int hf_modconv3x3_f16_pre_f32(float *out, const void *x_hi, const void *x_lo, const void *wt_hi,
                                         const void *wt_lo, int nterms, const float *d, const float *noise,
                                         const float *noise_w, long long noise_bstride, const float *bias, int batch,
                                         int cin, int cout, int h, int w, float alpha, float scale, float *rgb_raw,
                                         const float *rgb_wt, const float *rgb_s, void *split_hi, void *split_lo,
                                         const float *s_next, void *stream) {
  if ((!out && !rgb_raw && !split_hi) || !x_hi || !wt_hi || batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0 ||
      (noise && !noise_w) || (nterms != 1 && nterms != 3) || (nterms == 3 && !x_lo))
    return HF_E_INVALID;
  if (rgb_raw && (!rgb_wt || !rgb_s || (cout % 32))) return HF_E_INVALID;
  ConvParams P{};
  P.out = out; P.xh = x_hi; P.xl = x_lo; P.d = d; P.noise = noise; P.noise_w = noise_w; P.bias = bias;
  P.s_bstride = cin; P.d_bstride = cout;
  P.groups = 1;
  P.noise_bstride = noise_bstride;
  P.batch = batch; P.cin = cin; P.cout = cout; P.h = h; P.w = w; P.out_h = h; P.out_w = w; P.out_wv = w;
  P.stride = 1;
  P.act = bias ? ACT_LRELU : ACT_NONE;
  P.alpha = alpha; P.scale = scale;
  P.rgb_out = rgb_raw; P.rgb_w = rgb_wt; P.rgb_s = rgb_s;
  P.rgb_slabs = hf_modconv3x3_f16_rgb_slabs(cout);
  P.oh = split_hi; P.ol = split_lo; P.s_next = s_next;
  if (split_hi && ((cout & 7) || (nterms == 3 && !split_lo))) return HF_E_INVALID;
  return launch_conv_h(P, nterms, false, wt_hi, wt_lo, (hipStream_t)stream);
}

// The generator's LAST StyledConv with its ToRGB complete (ABI 13; the 1024^2 layer: 32 -> 32 channels, the row pipeline of
// convrow.hip): image = ToRGB's 1x1 modulated conv of the layer's output + rgb_bias + the x2-upsampled skip - no raw product
// in memory, no finishing launch.  HF_E_INVALID for every shape the row pipeline does not take (callers fall back to
// hf_modconv3x3_f16_pre_f32 + hf_torgb_f32: the same bits).
extern "C" int hf_modconv3x3_f16_pre_image_f32(float *image, const void *x_hi, const void *x_lo, const void *wt_hi,
                                               const void *wt_lo, int nterms, const float *d, const float *noise,
                                               const float *noise_w, long long noise_bstride, const float *bias, int batch,
                                               int cin, int cout, int h, int w, float alpha, float scale, const float *rgb_wt,
                                               const float *rgb_s, const float *rgb_bias, const float *skip,
                                               const float *kernel4x4, void *stream) {
  if (!image || !x_hi || !wt_hi || !rgb_wt || !rgb_s || !skip || !kernel4x4 || batch <= 0 || cin != 32 || cout != 32 || h <= 0 ||
      w <= 0 || (h & 1) || (w & 1) || (noise && !noise_w) || (nterms != 1 && nterms != 3) || (nterms == 3 && !x_lo) || !bias)
    return HF_E_INVALID;
  if (g_force_h || (g_h_tune & 16)) return HF_E_INVALID;  // the tiled form was asked for (tests, A/B)
  ConvParams P{};
  P.xh = x_hi; P.xl = x_lo; P.d = d; P.noise = noise; P.noise_w = noise_w; P.bias = bias;
  P.s_bstride = cin; P.d_bstride = cout;
  P.groups = 1;
  P.noise_bstride = noise_bstride;
  P.batch = batch; P.cin = cin; P.cout = cout; P.h = h; P.w = w; P.out_h = h; P.out_w = w; P.out_wv = w;
  P.stride = 1;
  P.act = ACT_LRELU;
  P.alpha = alpha; P.scale = scale;
  P.rgb_out = image; P.rgb_w = rgb_wt; P.rgb_s = rgb_s;
  P.rgb_slabs = 1;
  P.rgb_skip = skip; P.rgb_bias = rgb_bias; P.rgb_k4 = kernel4x4;
  return launch_conv_h(P, nterms, false, wt_hi, wt_lo, (hipStream_t)stream);
}

extern "C" int hf_modconv3x3_up_f16_pre_f32(float *tmp, const void *x_hi, const void *x_lo, const void *wt_hi,
                                            const void *wt_lo, int nterms, const float *d, int batch, int cin, int cout,
                                            int h, int w, int tmp_pitch, void *stream) {
  if (!tmp || !x_hi || !wt_hi || batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0 || tmp_pitch < 2 * w + 1 ||
      (nterms != 1 && nterms != 3) || (nterms == 3 && !x_lo))
    return HF_E_INVALID;
  ConvParams P{};
  P.out = tmp; P.xh = x_hi; P.xl = x_lo; P.d = d;
  P.s_bstride = cin; P.d_bstride = cout;
  P.groups = 1;
  P.batch = batch; P.cin = cin; P.cout = cout; P.h = h; P.w = w; P.out_h = 2 * h + 1; P.out_w = tmp_pitch;
  P.out_wv = 2 * w + 1;
  P.stride = 1;
  return launch_conv_h(P, nterms, true, wt_hi, wt_lo, (hipStream_t)stream);
}
