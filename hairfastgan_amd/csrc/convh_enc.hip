// convh_enc.hip - the encoders' 3x3 convolutions (stride 1 and 2) on the fp16 matrix cores.
//
// Reference operators: nn.Conv2d 3x3 with the surrounding inference-mode BatchNorm2d / PReLU /
// LeakyReLU / residual add of the IR-SE bottleneck (models/encoder4editing/models/encoders/
// helpers.py:93-120), the e4e style heads (psp_encoders.py:34-55, stride 2, grouped by family),
// IBasicBlock (models/FeatureStyleEncoder/arcface/iresnet.py:28-57) and the PostProcess trunk
// (models/Encoders.py:35-57).  Same contract as hf_conv2d_f32 (csrc/modconv.hip):
//   y = act( out_scale[co] * conv3x3_{stride, pad 1}( in_scale[ci]*x + in_shift[ci] ) + bias[co] ) + residual
// but the products run on v_mfma_f32_32x32x16_f16 - 16x the rate of the fp32 MFMA the round-1
// encoders used - in the operand modes of csrc/convh.hip: nterms 3 (fp32 operands split into fp16
// (hi, lo) pairs, hi*hi + hi*lo + lo*hi: fp32-class accuracy) or nterms 1 (operands rounded to fp16).
//
// Structure (one block per output tile; the generator's kernel in convh.hip additionally walks tiles
// persistently and takes pre-split input - encoder layers are too small for either to pay):
//   block tile = CT = 64 output channels x PT output pixels of ONE image (th rows x tw in {16, 32}
//   columns), waves = 2 (channels) x WAVES_PX, each 1 x 2 MFMA tiles;
//   K loop in stages of 16 input channels x 9 taps, double-buffered in LDS:
//     weights    [tap][kgroup][cout][8 halves] hi (+ lo) by LDS-DMA from the prepared layout of
//                hf_conv_split_weights_f16 (pre-scaled by 2^k, un-scaled in the epilogue),
//     activations one (halo pixel, kgroup) item per thread and step: 8 coalesced plane loads
//                (lanes = consecutive input columns), * in_scale + in_shift on REAL pixels (the zero
//                padding stays zero: the reason the BatchNorm in front of a conv cannot be folded
//                into its weights), saturating hi/lo split, one 16-byte LDS write per part;
//   STRIDE 2: the halo tile is (2 th + 1) x (2 tw + 1) input pixels, stored with its columns split by
//     parity ([row][parity][col/2]) so that the B fragment of every tap is again 32 consecutive
//     16-byte units (tap kx reads parity kx&1 from column px + (kx>>1));
//   epilogue = conv_common.h's store_tile (out scale, bias, PReLU / LeakyReLU, residual, groups).
// Grouped launches (blockIdx.y = group * co_tiles + co_tile) run the style heads of an e4e family
// level by level; every group's weights are a self-contained [9*cin*cout halves | 16-byte trailer].
#define HF_WANT_F16_SPLIT
#include "conv_common.h"
#include <type_traits>

using namespace hf_detail;

#ifndef HF_ENC_ILV
#define HF_ENC_ILV 1  // ping-pong K loop: the next tap's LDS fragment reads issued between the current tap's MFMAs (0: in front of them, A/B builds)
#endif
namespace {

// Scheduling pattern of one tap-step (HF_ENC_ILV): NR times (one MFMA, one LDS read), then the remaining MFMAs
template <int NR, int NM>
__device__ __forceinline__ void hf_enc_interleave() {
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
  }
  if constexpr (NM > NR) __builtin_amdgcn_sched_group_barrier(0x008, NM - NR, 0);
}


typedef _Float16 half8 __attribute__((ext_vector_type(8)));
constexpr int KH = 16;  // input channels per stage = K of one MFMA
#ifndef HF_ENC_S2MT
#define HF_ENC_S2MT 4  // pixel tiles per resident weight stage of the stride-2 multi-tile form (1 = off: A/B builds)
#endif
#ifndef HF_ENC_SWAP_XY
#define HF_ENC_SWAP_XY 1  // 0: always the (tiles, columns) grid (A/B builds)
#endif
#ifndef HF_ENC_PINGPONG
#define HF_ENC_PINGPONG 1  // 0: the one-phase K loop for every form (A/B builds: tools/build_variant.sh -DHF_ENC_PINGPONG=0)
#endif

// 16-byte units of one activation part per kgroup that the LDS tile is sized for
#ifndef HF_ENC_FAST_PROLOGUE
#define HF_ENC_FAST_PROLOGUE 1
#endif
#ifndef HF_ENC_PERSIST
#define HF_ENC_PERSIST 1
#endif
#ifdef HF_ENC_TRACE
// kernel-development build only (tools/probes/trace_enc_layer.py): where a block of conv_enc_h spends its time - s_memtime stamps of
// waves 0 and 7 in three blocks of the launch (the 1st, one of the 3rd round, one of a late round)
__device__ unsigned long long hf_enc_trace_buf[3 * 2 * 8];
#define HF_ENC_TRACE_POINT(id)                                                                                              \
  do {                                                                                                                      \
    const unsigned lin_ = persist ? (unsigned)lin : blockIdx.x + blockIdx.y * gridDim.x;                                        \
    const int sel_ = lin_ == 0 ? 0 : lin_ == 700 ? 1 : lin_ == 2000 ? 2 : -1;                                               \
    if (sel_ >= 0 && (threadIdx.x == 0 || threadIdx.x == 448))                                                              \
      hf_enc_trace_buf[(sel_ * 2 + (threadIdx.x ? 1 : 0)) * 8 + (id)] = __builtin_readcyclecounter();                       \
  } while (0)
#else
#define HF_ENC_TRACE_POINT(id) ((void)0)
#endif

template <int PT, int STRIDE>
constexpr int enc_npix() {
  // tw = 32: th = PT/32;  tw = 16: th = PT/16
  constexpr int a = (STRIDE == 1) ? (PT / 32 + 2) * (32 + 2) : (2 * (PT / 32) + 1) * 2 * (32 + 1);
  constexpr int b = (STRIDE == 1) ? (PT / 16 + 2) * (16 + 2) : (2 * (PT / 16) + 1) * 2 * (16 + 1);
  return a > b ? a : b;
}

// PRE: the activations arrive already affine-transformed, split into fp16 (hi, lo) and K-blocked
// ([image][cin/8][h][w][8 halves], hf_split_activation_f16; ConvParams::xh / xl): the halo tile is
// fetched by LDS-DMA like the weights - no per-element loads, no conversion (a shared input feeding
// many output-channel tiles / groups is then converted once instead of once per block).
// Wave tiles: WAVES_CO = 2, CT_TILES = 1 (each wave 32 channels x 64 pixels: small layers, many blocks) or
// WAVES_CO = 1, CT_TILES = 2 (each wave all 64 channels x 64 pixels, 2 x 2 MFMA tiles: 0.67 instead of 1 LDS
// fragment read per MFMA - the batched swap's layers, which fill the chip with 512-pixel tiles).
// VSPLIT: ConvParams::vsplit - the block walks all P.splits K slabs itself; at the end of a slab the slab's sum (what a real
// split-K block stores to its z slab) is added to a second accumulator set, slabs in z order from 0.0f like splitk_reduce, and
// the block's own epilogue finishes the tile: the bits of the two-launch form without its slabs and its second launch.
template <int NTERMS, int CT_TILES, bool VSPLIT>
__host__ __device__ constexpr bool enc_persist_ok() { return !(VSPLIT && CT_TILES == 2 && NTERMS == 3); }

template <int NTERMS, int PG, int WAVES_PX, int STRIDE, bool PRE, int CT_TILES = 1, int WAVES_CO = 2, bool VSPLIT = false>
__global__ __launch_bounds__(64 * WAVES_CO * WAVES_PX) void conv_enc_h(const ConvParams P, const _Float16 *__restrict__ wth_all,
                                                                       const _Float16 *__restrict__ wtl_all) {
  constexpr int NW = WAVES_CO * WAVES_PX;
  constexpr int NT = 64 * NW;
  constexpr int CT = 32 * CT_TILES * WAVES_CO;  // 64
  constexpr int PT = 32 * PG * WAVES_PX;
  constexpr int NPIX = enc_npix<PT, STRIDE>();
  constexpr int NPART = (NTERMS == 3) ? 2 : 1;
  constexpr int W_UNITS = 9 * 2 * CT;
  constexpr int X_UNITS = 2 * NPIX;
  constexpr int BUF_UNITS = NPART * (W_UNITS + X_UNITS);
  constexpr int N_WPIECE = NPART * W_UNITS / 64;
  constexpr int ND = (N_WPIECE + NW - 1) / NW;
  constexpr int XE = (X_UNITS + NT - 1) / NT;
  constexpr int OFF_WL = W_UNITS, OFF_XH = NPART * W_UNITS, OFF_XL = NPART * W_UNITS + X_UNITS;
  static_assert(XE <= 8, "conversion schedule: one item per tap-step");

  HF_DYN_LDS;
  half8 *lds = reinterpret_cast<half8 *>(hf_dyn_lds);  // [2][BUF_UNITS]
  float *sl = reinterpret_cast<float *>(lds + 2 * BUF_UNITS);  // in_scale [cin], in_shift [cin]
  const int cin4 = (P.cin + 3) & ~3;
  float *tl = sl + cin4;

  // Persistent form (ConvParams::persist; round 5): a block of the 512-pixel form is resident alone on its CU (158 KB of LDS), so
  // the ~5 us between the end of one block and the first instruction of the next - wave launch, LDS allocation, kernel arguments
  // - were spent with the CU empty, once per TILE: 12 times in a 64-channel layer of 430 us (tools/probes/trace_enc_layer.py:
  // 31 us per block inside, 36 us per block from outside).  The launch is now one resident block per CU that walks the plain
  // form's blocks b, b + persist, ... in their dispatch order (the same neighbours in L2 at any time); a tile's code is unchanged.
  // (not the 512-pixel form with the virtual split-K's second accumulator set: the loop costs it registers it does not have)
  const int persist = enc_persist_ok<NTERMS, CT_TILES, VSPLIT>() ? P.persist : 0;
  const int lin_n = persist ? P.lin_x * P.lin_y : 1;
  for (int lin = persist ? (int)blockIdx.x : 0; lin < lin_n; lin += persist ? persist : 1) {
  const int bx = persist ? lin % P.lin_x : (int)blockIdx.x, by = persist ? lin / P.lin_x : (int)blockIdx.y;
  HF_ENC_TRACE_POINT(0);
  const GroupOfs go = group_offsets(P, bx, by);
  const int grp = (P.groups > 1) ? (P.swap_xy ? bx : by) / P.co_tiles : 0;
  const long long wn = 9LL * P.cin * P.cout;
  const _Float16 *wth = wth_all + (long long)grp * (wn + 8);  // [weights | trailer] per group
  const _Float16 *wtl = wtl_all ? wtl_all + (long long)grp * wn : nullptr;
  const float w_unscale = *reinterpret_cast<const float *>(wth + wn);

  // (opaque per tile: what derives from the thread index is recomputed for every tile instead of being hoisted out of the tile
  // loop and kept in registers through the epilogue - the 512-pixel forms are at the 256-register limit there)
  int tid_ = threadIdx.x;
  if (enc_persist_ok<NTERMS, CT_TILES, VSPLIT>()) HF_OPAQUE_I32(tid_);
  const int tid = tid_, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wave_co = (wave / WAVES_PX) * 32 * CT_TILES;
  const int wave_pg = (wave % WAVES_PX) * PG;
  const int co0 = go.co_tile * CT;

  // ---- tile (uniform): OUTPUT pixels [ty0, ty0+th) x [tx0, tx0+tw) of image b0 ----
  const TileGeom G = P.g[0];
  int t = P.swap_xy ? by : bx;
  const int tx = t % G.tiles_x;
  t /= G.tiles_x;
  const int ty = t % G.tiles_y;
  const int b0 = t / G.tiles_y;
  const int tw = 1 << G.lg_tw, th = 1 << G.lg_th;
  const int ty0 = ty * th, tx0 = tx * tw;
  // halo tile: input rows STRIDE*ty0-1 .. (hp rows), columns STRIDE*tx0-1 .. (wcols columns);
  // LDS row pitch `wp` units; STRIDE 2: a row is [even columns: tw+1 units][odd columns: tw+1 units]
  const int hp = (th - 1) * STRIDE + 3, wcols = (tw - 1) * STRIDE + 3;
  const int wp2 = tw + 1;
  const int wp = (STRIDE == 1) ? wcols : 2 * wp2;
  const int n_items = hp * wcols;  // per kgroup, natural (row-major input) order: coalesced loads

  const long long plane = (long long)P.h * P.w;
  const int iplane = (int)plane;
  const float *xb = PRE ? nullptr : P.x + go.x + (long long)b0 * P.cin * plane;
  // PRE: image base of the split tensors; a per-group input ([groups][batch] images) when x_gstride != 0
  const long long img = (long long)(P.x_gstride ? grp * P.batch : 0) + b0;
  const char *xh_b = PRE ? static_cast<const char *>(P.xh) + img * (P.cin / 8) * plane * 16 : nullptr;
  const char *xl_b = (PRE && NTERMS == 3) ? static_cast<const char *>(P.xl) + img * (P.cin / 8) * plane * 16 : nullptr;

  const unsigned lds_addr0 = hf_lds_addr(lds);
  auto dma_piece = [&](int i, int chunk, int bufsel) {
    const int pc = wave + i * NW;
    if (pc < N_WPIECE) {
      const int part = pc / (W_UNITS / 64), q = pc % (W_UNITS / 64);
      const int u = q * 64 + lane;           // unit inside the part: (tap*2 + kg)*CT + co
      const int row = u / CT, col = u % CT;  // row = tap*2 + kg
      int off = (row * P.cout + co0 + col) * 16;
      HF_OPAQUE_I32(off);
      const _Float16 *src = (part ? wtl : wth) + (long long)chunk * 18 * P.cout * 8;
      hf_glds16_raw_s(src, (unsigned)off, lds_addr0 + (unsigned)(bufsel * BUF_UNITS + part * W_UNITS + q * 64) * 16u);
    }
  };
  // HF_ENC_FAST_PROLOGUE (round 5; tools/probes/trace_enc_layer.py: index arithmetic 2.4-3.3 k cycles, zero fill 1-1.5 k, first
  // stage 3-7.7 k - together 1.3 K stages per block, whatever the layer): the first stage's weights leave at once (they depend on
  // nothing below), its activations as soon as their addresses exist; only the halo units OUTSIDE the image are zeroed (by the
  // lanes that own them; the masked DMA never writes them) instead of both activation regions, and the divisions by the halo
  // pitch are divisions by its two possible values.
  const int c_first = (P.splits > 1 && !VSPLIT) ? (int)blockIdx.z * P.chunks_per_split : 0;
  constexpr bool FASTP = PRE && HF_ENC_FAST_PROLOGUE;
  if (FASTP) {
#pragma unroll
    for (int i = 0; i < ND; ++i) dma_piece(i, c_first, 0);
  }
  const bool tw32 = tw == 32;
  auto div_wp = [&](int u) { return STRIDE == 1 ? (tw32 ? u / 34 : u / 18) : (tw32 ? u / 66 : u / 34); };  // wp: tw + 2 | 2 (tw + 1)
  auto div_wp2 = [&](int r) { return tw32 ? r / 33 : r / 17; };                                              // wp2 = tw + 1
  auto div_wcols = [&](int v) { return STRIDE == 1 ? (tw32 ? v / 34 : v / 18) : (tw32 ? v / 65 : v / 33); };  // (tw - 1) STRIDE + 3
  // ---- per-thread staging items (stage invariant): plane offset of the input pixel (-1 zero fill,
  // -2 none) and the 16-byte LDS unit (incl. kgroup) it goes to.  Register staging enumerates the
  // halo in input order (coalesced loads); PRE enumerates LDS units (a DMA piece fills 64 consecutive
  // units, each lane fetching its own pixel's 16-byte unit) ----
  int e_src[XE], e_dst[XE], e_kg[XE];
#pragma unroll
  for (int e = 0; e < XE; ++e) {
    const int i = tid + e * NT;
    e_src[e] = -2;
    e_dst[e] = 0;
    if (!PRE) {
      const int kg = (i >= n_items) + (i >= 2 * n_items) + (i >= 3 * n_items), v = i - kg * n_items;  // (kg >= 2: no item)
      e_kg[e] = kg;
      if (kg < 2) {
        const int hr = div_wcols(v), hc = v - hr * wcols;
        const int ys = ty0 * STRIDE - 1 + hr, xc = tx0 * STRIDE - 1 + hc;
        e_src[e] = (ys >= 0 && ys < P.h && xc >= 0 && xc < P.w) ? ys * P.w + xc : -1;
        e_dst[e] = kg * NPIX + ((STRIDE == 1) ? hr * wp + hc : hr * wp + (hc & 1) * wp2 + (hc >> 1));
      }
    } else {
      const int kg = i / NPIX, u = i - kg * NPIX;
      e_kg[e] = kg;
      e_dst[e] = i;
      if (i < X_UNITS && u < hp * wp) {
        const int hr = div_wp(u), r = u - hr * wp;
        const int rq = div_wp2(r);
        const int hc = (STRIDE == 1) ? r : 2 * (r - rq * wp2) + rq;
        if (hc < wcols) {
          const int ys = ty0 * STRIDE - 1 + hr, xc = tx0 * STRIDE - 1 + hc;
          e_src[e] = (ys >= 0 && ys < P.h && xc >= 0 && xc < P.w) ? ys * P.w + xc : -1;
        }
      }
    }
  }
  if (!PRE)
    for (int i = tid; i < P.cin; i += NT) {
      sl[i] = P.s ? P.s[i] : 1.0f;
      tl[i] = P.t ? P.t[i] : 0.0f;
    }

  // PRE: item e of stage `chunk` straight into LDS (units of pixels outside the image were zeroed once)
  auto dma_x = [&](int e, int chunk, int bufsel) {
    const int i = tid + e * NT;
    if (i - lane >= X_UNITS) return;  // whole piece beyond the part (uniform per wave)
    const bool inside = e_src[e] >= 0;
    int off = inside ? (e_kg[e] * iplane + e_src[e]) * 16 : 0;
    HF_OPAQUE_I32(off);
    const long long cofs = (long long)chunk * 2 * plane * 16;  // 2 channel blocks per stage
    const unsigned dst = lds_addr0 + (unsigned)(bufsel * BUF_UNITS + OFF_XH + (i - lane)) * 16u;
    hf_glds16_raw_s_if(inside, xh_b + cofs, (unsigned)off, dst);
    if (NTERMS == 3) hf_glds16_raw_s_if(inside, xl_b + cofs, (unsigned)off, dst + (unsigned)X_UNITS * 16u);
  };
  float xr[XE][8];
  auto load_item = [&](int e, int chunk) {
    const char *xc = reinterpret_cast<const char *>(xb + (long long)chunk * KH * plane);
    int off = (e_src[e] >= 0) ? e_kg[e] * 8 * iplane + e_src[e] : 0;
    HF_OPAQUE_I32(off);
#pragma unroll
    for (int k = 0; k < 8; ++k) xr[e][k] = *reinterpret_cast<const float *>(xc + (unsigned)((off + k * iplane) * 4));
  };
  auto convert_item = [&](int e, int chunk, half8 *buf) {
    if (e_src[e] == -2) return;
    half8 hi, lo;
    bool ovf = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int ci = chunk * KH + e_kg[e] * 8 + k;
      const float v = (e_src[e] >= 0) ? fmaf(xr[e][k], sl[ci], tl[ci]) : 0.0f;
      _Float16 hv, lv;
      hf_split_f16(v, hv, lv, ovf);
      hi[k] = hv;
      lo[k] = lv;
    }
    hf_note_overflow(ovf);
    buf[OFF_XH + e_dst[e]] = hi;
    if (NTERMS == 3) buf[OFF_XL + e_dst[e]] = lo;
  };

  f32x16 acc[1][CT_TILES][PG];
#pragma unroll
  for (int ct = 0; ct < CT_TILES; ++ct)
#pragma unroll
    for (int g = 0; g < PG; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][ct][g][r] = 0.0f;

  // LDS unit (inside a kgroup) of the lane's pixel of group g at tap (0, 0)
  int pix0[PG];
#pragma unroll
  for (int g = 0; g < PG; ++g) {
    const int p = (wave_pg + g) * 32 + li;
    const int py = (p >> G.lg_tw) & (th - 1), px = p & (tw - 1);
    pix0[g] = py * STRIDE * wp + px;
  }

  const int nchunks_all = P.cin / KH;
  const int c_begin = (P.splits > 1 && !VSPLIT) ? (int)blockIdx.z * P.chunks_per_split : 0;
  const int c_end = (P.splits > 1 && !VSPLIT) ? min(nchunks_all, c_begin + P.chunks_per_split) : nchunks_all;
  f32x16 vsum[VSPLIT ? CT_TILES : 1][VSPLIT ? PG : 1];  // VSPLIT: the slabs added so far
  int slab_left = P.chunks_per_split;                   // VSPLIT: chunks until the current slab ends
  if (VSPLIT) {
#pragma unroll
    for (int ct = 0; ct < CT_TILES; ++ct)
#pragma unroll
      for (int g = 0; g < PG; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) vsum[VSPLIT ? ct : 0][VSPLIT ? g : 0][r] = 0.0f;
  }
  HF_ENC_TRACE_POINT(1);  // index arithmetic done
  if (FASTP) {
#pragma unroll
    for (int e = 0; e < XE; ++e) dma_x(e, c_begin, 0);
    half8 z;
#pragma unroll
    for (int k = 0; k < 8; ++k) z[k] = (_Float16)0.0f;
#pragma unroll
    for (int e = 0; e < XE; ++e)
      if (e_src[e] == -1) {  // a halo unit outside the image: zero in both buffers, for every stage
        lds[OFF_XH + e_dst[e]] = z;
        lds[BUF_UNITS + OFF_XH + e_dst[e]] = z;
        if (NTERMS == 3) {
          lds[OFF_XL + e_dst[e]] = z;
          lds[BUF_UNITS + OFF_XL + e_dst[e]] = z;
        }
      }
  } else if (PRE) {  // zero the activation regions of both buffers once: DMA-masked units (padding) stay zero
    half8 z;
#pragma unroll
    for (int k = 0; k < 8; ++k) z[k] = (_Float16)0.0f;
    for (int i = tid; i < NPART * X_UNITS; i += NT) {
      lds[OFF_XH + i] = z;
      lds[BUF_UNITS + OFF_XH + i] = z;
    }
  }
  if (!FASTP) __syncthreads();  // sl / tl (or the zero fill) visible
  HF_ENC_TRACE_POINT(2);  // LDS zero fill done
  if (!FASTP) {
#pragma unroll
    for (int i = 0; i < ND; ++i) dma_piece(i, c_begin, 0);
  }
  if (FASTP) {
  } else if (PRE) {
#pragma unroll
    for (int e = 0; e < XE; ++e) dma_x(e, c_begin, 0);
  } else {
#pragma unroll
    for (int e = 0; e < XE; ++e) load_item(e, c_begin);
#pragma unroll
    for (int e = 0; e < XE; ++e) convert_item(e, c_begin, lds);
  }
  hf_barrier_keep_young<0>();
  HF_ENC_TRACE_POINT(3);  // first stage landed

  // ---- K loop.  PP ("ping-pong", pre-split input, eight waves = two per SIMD): the two waves of a SIMD (w and w + 4: a
  // workgroup's waves go round the four SIMDs) take TURNS on the matrix pipe - phase A: the first half of the block runs its
  // 108 MFMAs of the stage back to back while the second half issues its LDS-DMA copies of the next stage; s_barrier; phase B:
  // roles swapped; s_barrier.  In the one-phase form (every wave: a tap's side work, then its MFMAs, all eight in lock-step)
  // the pipe idles while both waves of a SIMD sit in their DMA issues (100-185 cycles each, in-order waves: a tap-step took
  // 1.2-1.5 k cycles for 768 cycles of MFMA time, profiles/r04a_trace_fused.txt); here the partner's MFMAs cover them
  // (MI355X_MICROARCH.md, "Two waves per SIMD": matrix beside memory).  Same K order per wave: equal bits.
  constexpr bool PP = PRE && NW == 8 && HF_ENC_PINGPONG;
  const int half = wave >> 2;  // PP: 0 computes in phase A, 1 in phase B
  half8 ah[2][CT_TILES], al[2][CT_TILES], bh[2][PG], bl[2][PG];
  auto compute_stage = [&](const half8 *buf) {  // the 9 taps of one K stage: fragments of tap+1 fetched under the MFMAs of tap
    const half8 *a_hi = buf + lh * CT + wave_co + li;  // + tap*2*CT
    const half8 *b_hi = buf + OFF_XH + lh * NPIX;
    auto fetch = [&](int slot, int tap) {
      const int ky = tap / 3, kx = tap % 3;
      const int toff = (STRIDE == 1) ? ky * wp + kx : ky * wp + (kx & 1) * wp2 + (kx >> 1);
#pragma unroll
      for (int ct = 0; ct < CT_TILES; ++ct) {
        ah[slot][ct] = a_hi[tap * 2 * CT + ct * 32];
        if (NTERMS == 3) al[slot][ct] = a_hi[OFF_WL + tap * 2 * CT + ct * 32];
      }
#pragma unroll
      for (int g = 0; g < PG; ++g) {
        bh[slot][g] = b_hi[pix0[g] + toff];
        if (NTERMS == 3) bl[slot][g] = b_hi[X_UNITS + pix0[g] + toff];
      }
    };
    fetch(0, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int s_ = tap & 1;
      if (tap + 1 < 9) fetch(s_ ^ 1, tap + 1);
      if (!HF_ENC_ILV) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ct = 0; ct < CT_TILES; ++ct)
#pragma unroll
        for (int g = 0; g < PG; ++g)
          acc[0][ct][g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s_][ct], bh[s_][g], acc[0][ct][g], 0, 0, 0);
      if (NTERMS == 3) {
#pragma unroll
        for (int ct = 0; ct < CT_TILES; ++ct)
#pragma unroll
          for (int g = 0; g < PG; ++g)
            acc[0][ct][g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s_][ct], bl[s_][g], acc[0][ct][g], 0, 0, 0);
#pragma unroll
        for (int ct = 0; ct < CT_TILES; ++ct)
#pragma unroll
          for (int g = 0; g < PG; ++g)
            acc[0][ct][g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s_][ct], bh[s_][g], acc[0][ct][g], 0, 0, 0);
      }
      if constexpr (HF_ENC_ILV != 0) {
        // (round 6) the next tap's fragment reads BETWEEN this tap's MFMAs, one behind each: in its turn on the pipe the wave is
        // alone on its SIMD - nothing else covers the time the reads take to issue (see HF_H_ILV, csrc/convh.hip)
        constexpr int NRD = (CT_TILES + PG) * (NTERMS == 3 ? 2 : 1), NM = CT_TILES * PG * NTERMS;
        if (tap + 1 < 9) hf_enc_interleave<(NRD < NM ? NRD : NM), NM>();
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto stage_dma = [&](int chunk, int bufsel) {  // PP: this wave's share of a stage's copies, issued back to back
#pragma unroll
    for (int i = 0; i < ND; ++i) dma_piece(i, chunk, bufsel);
#pragma unroll
    for (int e = 0; e < XE; ++e) dma_x(e, chunk, bufsel);
  };
  auto slab_end = [&](bool more) {  // VSPLIT: a slab ends - its sum (pre-scale undone, exact) joins the earlier slabs'
    if (VSPLIT && (--slab_left == 0 || !more)) {
      slab_left = P.chunks_per_split;
#pragma unroll
      for (int ct = 0; ct < CT_TILES; ++ct)
#pragma unroll
        for (int g = 0; g < PG; ++g) {
          vsum[VSPLIT ? ct : 0][VSPLIT ? g : 0] += acc[0][ct][g] * w_unscale;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[0][ct][g][r] = 0.0f;
        }
    }
  };

  for (int c = c_begin; c < c_end; ++c) {
    const int cb = (c - c_begin) & 1;
    half8 *buf = lds + cb * BUF_UNITS, *nbuf = lds + (cb ^ 1) * BUF_UNITS;
    const bool more = c + 1 < c_end;
    if constexpr (PP) {
      // Hazards: stage c is complete since the end-of-stage barrier of stage c-1 (every wave drained its copies: vmcnt(0)); the
      // copies of stage c+1 go to the buffer stage c-1 was read from, whose last readers (second half, phase B of c-1) passed
      // that barrier too.  The mid-stage barrier only swaps the roles (copies stay in flight across it).  The last stage
      // has no end barrier: the first half's epilogue runs under the second half's MFMAs.
      if (half == 0) {
        compute_stage(buf);
        slab_end(more);
        hf_barrier_lds();
        if (more) {
          stage_dma(c + 1, cb ^ 1);
          hf_barrier_keep_young<0>();
        }
      } else {
        if (more) stage_dma(c + 1, cb ^ 1);
        hf_barrier_lds();
        compute_stage(buf);
        slab_end(more);
        if (more) hf_barrier_keep_young<0>();
      }
      continue;
    }
    const half8 *a_hi = buf + lh * CT + wave_co + li;  // + tap*2*CT
    const half8 *b_hi = buf + OFF_XH + lh * NPIX;
    auto fetch = [&](int slot, int tap) {
      const int ky = tap / 3, kx = tap % 3;
      const int toff = (STRIDE == 1) ? ky * wp + kx : ky * wp + (kx & 1) * wp2 + (kx >> 1);
#pragma unroll
      for (int ct = 0; ct < CT_TILES; ++ct) {
        ah[slot][ct] = a_hi[tap * 2 * CT + ct * 32];
        if (NTERMS == 3) al[slot][ct] = a_hi[OFF_WL + tap * 2 * CT + ct * 32];
      }
#pragma unroll
      for (int g = 0; g < PG; ++g) {
        bh[slot][g] = b_hi[pix0[g] + toff];
        if (NTERMS == 3) bl[slot][g] = b_hi[X_UNITS + pix0[g] + toff];
      }
    };
    fetch(0, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int s_ = tap & 1;
      if (tap + 1 < 9) fetch(s_ ^ 1, tap + 1);
      if (more) {
        if (!PRE && tap == 0) {
#pragma unroll
          for (int e = 0; e < XE; ++e) load_item(e, c + 1);
        }
#pragma unroll
        for (int i = 0; i < ND; ++i)
          if (i / ((ND + 2) / 3) == tap) dma_piece(i, c + 1, cb ^ 1);
        if (PRE) {
#pragma unroll
          for (int e = 0; e < XE; ++e)
            if (e == tap) dma_x(e, c + 1, cb ^ 1);
        } else if (tap >= 9 - XE) {
          convert_item(tap - (9 - XE), c + 1, nbuf);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ct = 0; ct < CT_TILES; ++ct)
#pragma unroll
        for (int g = 0; g < PG; ++g)
          acc[0][ct][g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s_][ct], bh[s_][g], acc[0][ct][g], 0, 0, 0);
      if (NTERMS == 3) {
#pragma unroll
        for (int ct = 0; ct < CT_TILES; ++ct)
#pragma unroll
          for (int g = 0; g < PG; ++g)
            acc[0][ct][g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s_][ct], bl[s_][g], acc[0][ct][g], 0, 0, 0);
#pragma unroll
        for (int ct = 0; ct < CT_TILES; ++ct)
#pragma unroll
          for (int g = 0; g < PG; ++g)
            acc[0][ct][g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s_][ct], bh[s_][g], acc[0][ct][g], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    slab_end(more);
    hf_barrier_keep_young<0>();  // next stage complete (DMA landed, conversions written), current one free
  }

  HF_ENC_TRACE_POINT(4);  // K loop done
  // undo the weights' power-of-two pre-scale (exact), then the shared epilogue
#pragma unroll
  for (int ct = 0; ct < CT_TILES; ++ct)
#pragma unroll
    for (int g = 0; g < PG; ++g) {
      if (VSPLIT) acc[0][ct][g] = vsum[VSPLIT ? ct : 0][VSPLIT ? g : 0];
      else
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][ct][g][r] *= w_unscale;
    }
  if (P.d_bstride == 0 && !P.noise)
    store_tile_rows<CT_TILES, PG>(P, G, go, acc, co0 + wave_co, wave_pg, li, lh, ty0, tx0, b0);
  else
    store_tile<CT_TILES, PG, false>(P, G, go, acc, co0 + wave_co, wave_pg, li, lh, ty0, tx0, b0);
  HF_ENC_TRACE_POINT(5);  // epilogue issued
#ifdef HF_ENC_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  HF_ENC_TRACE_POINT(6);  // stores acknowledged
#endif
  // split-K without a second launch: the tile's last block adds the slabs and runs the epilogue (conv_common.h)
  if (!VSPLIT && P.splits > 1 && P.counters && splitk_arrive_last(P, reinterpret_cast<int *>(hf_dyn_lds)))
    store_tile<CT_TILES, PG, false, 1, true>(P, G, go, acc, co0 + wave_co, wave_pg, li, lh, ty0, tx0, b0);
  if (persist) __syncthreads();  // every wave is done with the stage buffers (and sl / tl): the next tile's copies may land
  }
}

// ---- stride 2, several pixel tiles per resident weight stage (round 5) -------------------------------------------------------
// A stride-2 conv reads four input pixels per output pixel: its 64 co x 128 px blocks move 36.9 KB of weights + 38 KB of
// activations per 216 MFMAs - four times the bytes per MFMA of the stride-1 512-pixel form - and are bound by that L2 -> LDS
// traffic (8.2 TB/s = 16 B/clk/CU on the e4e heads' first level).  A 256-pixel tile does not fit LDS twice; what does fit is
// to keep a K stage's WEIGHTS for MT consecutive 128-pixel tiles: the block holds MT accumulator tiles per wave (16 registers
// each), walks step (chunk c, tile j) with the activations of the next step and a share of the next chunk's weights arriving by
// LDS-DMA meanwhile (two activation buffers by step parity, two weight buffers by chunk parity: the same 150 KB), one barrier per
// step as before.  Bytes per MFMA: (38 + 36.9 / MT) KB per 216 instead of 75.  Per output the K order is that of conv_enc_h:
// equal bits (tests/test_sim_encoders.py).  Pre-split input only; real split-K launches keep the one-tile form.
#ifndef HF_ENC_S2MT_AREG
#define HF_ENC_S2MT_AREG 1
#endif
template <int NTERMS, int MT, bool VSPLIT>
__global__ __launch_bounds__(512) void conv_enc_s2mt_h(const ConvParams P, const _Float16 *__restrict__ wth_all,
                                                       const _Float16 *__restrict__ wtl_all) {
  constexpr int WAVES_PX = 4, NW = 8, NT = 64 * NW, CT = 64, PT = 128;
  constexpr int NPIX = enc_npix<PT, 2>();
  constexpr int NPART = (NTERMS == 3) ? 2 : 1;
  constexpr int W_UNITS = 9 * 2 * CT, X_UNITS = 2 * NPIX;
  constexpr int BUF_UNITS = NPART * (W_UNITS + X_UNITS);
  constexpr int N_WPIECE = NPART * W_UNITS / 64;
  constexpr int ND = (N_WPIECE + NW - 1) / NW;
  constexpr int XE = (X_UNITS + NT - 1) / NT;
  constexpr int OFF_WL = W_UNITS, OFF_XH = NPART * W_UNITS, OFF_XL = NPART * W_UNITS + X_UNITS;

  HF_DYN_LDS;
  half8 *lds = reinterpret_cast<half8 *>(hf_dyn_lds);  // [2][BUF_UNITS]: W of chunk parity b | X of step parity b

  const GroupOfs go = group_offsets(P);
  const int grp = (P.groups > 1) ? (P.swap_xy ? (int)blockIdx.x : (int)blockIdx.y) / P.co_tiles : 0;
  const long long wn = 9LL * P.cin * P.cout;
  const _Float16 *wth = wth_all + (long long)grp * (wn + 8);
  const _Float16 *wtl = wtl_all ? wtl_all + (long long)grp * wn : nullptr;
  const float w_unscale = *reinterpret_cast<const float *>(wth + wn);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wave_co = (wave / WAVES_PX) * 32;
  const int wave_pg = wave % WAVES_PX;
  const int co0 = go.co_tile * CT;

  const TileGeom G = P.g[0];
  const int tw = 1 << G.lg_tw, th = 1 << G.lg_th;
  const int hp = (th - 1) * 2 + 3, wcols = (tw - 1) * 2 + 3;
  const int wp2 = tw + 1, wp = 2 * wp2;
  const int n_tiles = G.tiles_x * G.tiles_y * G.tiles_b;
  const int t_first = (P.swap_xy ? (int)blockIdx.y : (int)blockIdx.x) * MT;
  const long long plane = (long long)P.h * P.w;
  const int iplane = (int)plane;

  // the MT tiles of this block: origin, image and - per staging item - the input pixel of its 16-byte unit
  int tx0[MT], ty0[MT], b0[MT], e_src[MT][XE];
  bool valid[MT];
  const char *xh_b[MT], *xl_b[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    int t = t_first + j;
    valid[j] = t < n_tiles;
    if (!valid[j]) t = n_tiles - 1;  // (computes the last tile again, stores nothing: uniform barriers)
    const int tx = t % G.tiles_x;
    t /= G.tiles_x;
    const int ty = t % G.tiles_y;
    b0[j] = t / G.tiles_y;
    ty0[j] = ty * th;
    tx0[j] = tx * tw;
    const long long img = (long long)(P.x_gstride ? grp * P.batch : 0) + b0[j];
    xh_b[j] = static_cast<const char *>(P.xh) + img * (P.cin / 8) * plane * 16;
    xl_b[j] = (NTERMS == 3) ? static_cast<const char *>(P.xl) + img * (P.cin / 8) * plane * 16 : nullptr;
#pragma unroll
    for (int e = 0; e < XE; ++e) {
      const int i = tid + e * NT;
      const int kg = i / NPIX, u = i - kg * NPIX;
      e_src[j][e] = -2;
      if (i < X_UNITS && u < hp * wp) {
        const int hr = (tw == 32) ? u / 66 : u / 34, r = u - hr * wp;  // wp = 2 (tw + 1), tw 16 | 32 (launch_enc)
        const int rq = (tw == 32) ? r / 33 : r / 17;
        const int hc = 2 * (r - rq * wp2) + rq;
        if (hc < wcols) {
          const int ys = ty0[j] * 2 - 1 + hr, xc = tx0[j] * 2 - 1 + hc;
          e_src[j][e] = (ys >= 0 && ys < P.h && xc >= 0 && xc < P.w) ? ys * P.w + xc : -1;
        }
      }
    }
  }

  const unsigned lds_addr0 = hf_lds_addr(lds);
  auto dma_piece = [&](int i, int chunk, int wbuf) {
    const int pc = wave + i * NW;
    if (pc < N_WPIECE) {
      const int part = pc / (W_UNITS / 64), q = pc % (W_UNITS / 64);
      const int u = q * 64 + lane;
      const int row = u / CT, col = u % CT;
      int off = (row * P.cout + co0 + col) * 16;
      HF_OPAQUE_I32(off);
      const _Float16 *src = (part ? wtl : wth) + (long long)chunk * 18 * P.cout * 8;
      hf_glds16_raw_s(src, (unsigned)off, lds_addr0 + (unsigned)(wbuf * BUF_UNITS + part * W_UNITS + q * 64) * 16u);
    }
  };
  // activations of (chunk, tile j) into the X region of buffer xbuf; halo units outside the image are written as zeros by
  // their lanes (the tiles of a block differ in which units those are: nothing can be zeroed once and for all)
  auto dma_x = [&](int j, int e, int chunk, int xbuf) {
    const int i = tid + e * NT;
    if (i - lane >= X_UNITS) return;
    const int kg = i / NPIX;
    const bool inside = e_src[j][e] >= 0;
    int off = inside ? (kg * iplane + e_src[j][e]) * 16 : 0;
    HF_OPAQUE_I32(off);
    const long long cofs = (long long)chunk * 2 * plane * 16;
    const unsigned dst = lds_addr0 + (unsigned)(xbuf * BUF_UNITS + OFF_XH + (i - lane)) * 16u;
    hf_glds16_raw_s_if(inside, xh_b[j] + cofs, (unsigned)off, dst);
    if (NTERMS == 3) hf_glds16_raw_s_if(inside, xl_b[j] + cofs, (unsigned)off, dst + (unsigned)X_UNITS * 16u);
    if (e_src[j][e] == -1) {
      half8 z;
#pragma unroll
      for (int k = 0; k < 8; ++k) z[k] = (_Float16)0.0f;
      lds[xbuf * BUF_UNITS + OFF_XH + i] = z;
      if (NTERMS == 3) lds[xbuf * BUF_UNITS + OFF_XL + i] = z;
    }
  };

  f32x16 acc[MT][1][1][1];
  f32x16 vsum[VSPLIT ? MT : 1];
#pragma unroll
  for (int j = 0; j < MT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc[j][0][0][0][r] = 0.0f;
      if (VSPLIT) vsum[VSPLIT ? j : 0][r] = 0.0f;
    }
  int slab_left = P.chunks_per_split;

  const int p = wave_pg * 32 + li;
  const int pix0 = ((p >> G.lg_tw) & (th - 1)) * 2 * wp + (p & (tw - 1));
  const int nchunks = P.cin / KH;

  // prologue: weights of chunk 0, activations of step (0, tile 0)
#pragma unroll
  for (int i = 0; i < ND; ++i) dma_piece(i, 0, 0);
#pragma unroll
  for (int e = 0; e < XE; ++e) dma_x(0, e, 0, 0);
  hf_barrier_keep_young<0>();

  // HF_ENC_S2MT_AREG: the chunk's weight fragments (9 taps x hi / lo, 72 registers) are read from LDS during the chunk's first
  // tile and stay in registers for its other MT - 1 tiles: 36 -> 18 ds_read_b128 per tile step for those (the 128-pixel form has
  // ONE accumulator tile per wave, 1.33 LDS reads per MFMA - twice the 512-pixel form's - and 8 waves' reads take longer than
  // their MFMAs)
  // (not with four tiles AND the virtual split-K's second accumulator set: 256 registers + scratch)
  constexpr bool AREG = HF_ENC_S2MT_AREG && !(VSPLIT && MT == 4 && NTERMS == 3);
  half8 wh[AREG ? 9 : 2], wl[AREG ? 9 : 2], bh[2], bl[2];
  int step = 0;
  for (int c = 0; c < nchunks; ++c) {
    const int wb = c & 1;
    const bool more_c = c + 1 < nchunks;
#pragma unroll
    for (int j = 0; j < MT; ++j, ++step) {
      const int xb = step & 1;
      // the next step's activations, a share of the next chunk's weights
      if (j + 1 < MT) {
#pragma unroll
        for (int e = 0; e < XE; ++e) dma_x(j + 1 < MT ? j + 1 : 0, e, c, xb ^ 1);
      } else if (more_c) {
#pragma unroll
        for (int e = 0; e < XE; ++e) dma_x(0, e, c + 1, xb ^ 1);
      }
      if (more_c) {
#pragma unroll
        for (int i = 0; i < ND; ++i)
          if (i % MT == j) dma_piece(i, c + 1, wb ^ 1);
      }
      const half8 *a_hi = lds + wb * BUF_UNITS + lh * CT + wave_co + li;
      const half8 *b_hi = lds + xb * BUF_UNITS + OFF_XH + lh * NPIX;
      const bool load_w = !AREG || j == 0;
      auto fetch = [&](int slot, int tap) {
        const int ky = tap / 3, kx = tap % 3;
        const int toff = ky * wp + (kx & 1) * wp2 + (kx >> 1);
        if (load_w) {
          const int ws = AREG ? tap : slot;
          wh[ws] = a_hi[tap * 2 * CT];
          if (NTERMS == 3) wl[ws] = a_hi[OFF_WL + tap * 2 * CT];
        }
        bh[slot] = b_hi[pix0 + toff];
        if (NTERMS == 3) bl[slot] = b_hi[X_UNITS + pix0 + toff];
      };
      fetch(0, 0);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int s_ = tap & 1;
        const int ws = AREG ? tap : s_;
        if (tap + 1 < 9) fetch(s_ ^ 1, tap + 1);
        __builtin_amdgcn_sched_barrier(0);
        acc[j][0][0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ws], bh[s_], acc[j][0][0][0], 0, 0, 0);
        if (NTERMS == 3) {
          acc[j][0][0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ws], bl[s_], acc[j][0][0][0], 0, 0, 0);
          acc[j][0][0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[ws], bh[s_], acc[j][0][0][0], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      hf_barrier_keep_young<0>();  // next step's copies landed, this step's buffers free
    }
    if (VSPLIT && (--slab_left == 0 || !more_c)) {  // a slab ends (see conv_enc_h)
      slab_left = P.chunks_per_split;
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        vsum[VSPLIT ? j : 0] += acc[j][0][0][0] * w_unscale;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][0][0][0][r] = 0.0f;
      }
    }
  }
  // (one epilogue per tile with a constant index: a loop the compiler does not unroll puts the per-tile arrays into scratch)
  auto finish = [&](auto jc) {
    constexpr int j = decltype(jc)::value;
    if (VSPLIT) acc[j][0][0][0] = vsum[VSPLIT ? j : 0];
    else
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][0][0][0][r] *= w_unscale;
    if (valid[j]) store_tile_rows<1, 1>(P, G, go, acc[j], co0 + wave_co, wave_pg, li, lh, ty0[j], tx0[j], b0[j]);
  };
  finish(std::integral_constant<int, 0>{});
  finish(std::integral_constant<int, 1>{});
  if constexpr (MT == 4) {
    finish(std::integral_constant<int, 2>{});
    finish(std::integral_constant<int, 3>{});
  }
  static_assert(MT == 2 || MT == 4, "tiles per resident weight stage");
}

// split-K plan.  The tile forms that small launches take hold 100-117 KB of LDS: ONE block per CU, so a launch runs in
// ceil(blocks / 256) rounds of resident blocks and a K split only pays while it does not add a round.  Cost model (us; round 5,
// from the batch-3 layers of a single swap - a block's fixed cost of ~3 us: LDS zero fill, first stage's DMA latency, epilogue;
// ~0.9 us per 16-channel K stage; ~6 us for the second pass incl. its launch):
//   T(s) = ceil(blocks * s / 256) * (3 + 0.9 * ceil(nchunks / s)) + (s > 1 ? 6 : 0),   s <= nchunks / 4, s <= 16
// e.g. 96 blocks x 16 stages (256 -> 256 @ 32^2, batch 3): s = 2 (16.2; s = 1: 17.4, the old plan's s = 4 - 384 blocks, two rounds -
// 19.2); 48 blocks x 32 stages (512 @ 16^2): s = 5; 192 blocks x 8 stages (128 @ 64^2): no split (the old plan: 2).
// -DHF_ENC_PLAN_OLD: the round-2 rule (>= 1.5 blocks per CU) for A/B builds.
inline int enc_splitk_plan(long long blocks, int nchunks) {
  if (blocks >= 256 || nchunks < 8) return 1;
#ifdef HF_ENC_PLAN_OLD
  int s = (int)((384 + blocks - 1) / blocks);
  if (s > nchunks / 4) s = nchunks / 4;
  if (s > 16) s = 16;
  return s < 2 ? 1 : s;
#else
  int best = 1;
  double best_t = 1e30;
  const int smax = nchunks / 4 < 16 ? nchunks / 4 : 16;
  for (int s = 1; s <= smax; ++s) {
    const double rounds = (double)((blocks * s + 255) / 256);
    const double t = rounds * (3.0 + 0.9 * ((nchunks + s - 1) / s)) + (s > 1 ? 6.0 : 0.0);
    if (t < best_t - 1e-9) {
      best_t = t;
      best = s;
    }
  }
  return best;
#endif
}

// force_splits > 0 (batch-invariant plans): the K partition is given - the canonical plan of run_enc - and only its
// execution is decided here: one block per slab (grid.z, partial slabs + splitk_reduce) when the output grid alone leaves
// the chip empty, else ConvParams::vsplit (every block walks all slabs; same bits).
thread_local int g_s2mt_last = 0;  // 4 / 2: the last stride-2 launch took the multi-tile form with that many tiles (hf_debug_last_path 605 / 606)

template <int NTERMS, int PG, int WAVES_PX, int STRIDE, int CT_TILES = 1, int WAVES_CO = 2>
int launch_enc(ConvParams &P, const _Float16 *wth, const _Float16 *wtl, float *workspace, long long workspace_floats,
               hipStream_t st, bool plan_only = false, int force_splits = 0) {
  constexpr int CT = 32 * CT_TILES * WAVES_CO, NT = 64 * WAVES_CO * WAVES_PX;
  static_assert(CT == 64, "64 output channels per block");
  constexpr int PT = 32 * PG * WAVES_PX;
  constexpr int NPIX = enc_npix<PT, STRIDE>();
  constexpr int NPART = (NTERMS == 3) ? 2 : 1;
  if (P.cin % KH || P.cout % CT || P.stride != STRIDE) return HF_E_INVALID;
  if ((long long)P.cin * P.h * P.w >= (1LL << 31)) return HF_E_INVALID;
  P.n_geom = 1;
  P.g[0] = make_geom(0, 0, P.out_h, P.out_w, P.batch, PT, 0);
  const TileGeom &G = P.g[0];
  const int tw = 1 << G.lg_tw, th = 1 << G.lg_th;
  if (G.lg_nb != 0 || tw * th != PT || (tw != 16 && tw != 32)) return HF_E_INVALID;  // one image per tile, 16 / 32 columns
  const int hp = (th - 1) * STRIDE + 3, wcols = (tw - 1) * STRIDE + 3;
  const int units = (STRIDE == 1) ? hp * wcols : hp * 2 * (tw + 1);
  if (units > NPIX || 2 * hp * wcols > 8 * NT) return HF_E_INVALID;
  P.co_tiles = P.cout / CT;
  const int groups = P.groups > 1 ? P.groups : 1;
  const long long blocks = (long long)geom_blocks(G) * P.co_tiles * groups;
  // what this launch needs to fill the chip by itself (hf_debug_set_tuning bits 24-31: tests lower the block count from which a
  // launch counts as chip-filling, so that small shapes reach the virtual split)
  const int fill = (g_h_tune >> 24) & 255;
  const int own_splits = (fill && blocks >= fill) ? 1 : enc_splitk_plan(blocks, P.cin / KH);
  P.splits = force_splits > 0 ? force_splits : own_splits;
  P.chunks_per_split = hf_cdiv(P.cin / KH, P.splits);
  P.splits = hf_cdiv(P.cin / KH, P.chunks_per_split);  // no empty split
  P.zslab = (long long)groups * P.batch * P.cout * P.out_h * P.out_w;
  // virtual split: the encoder-type epilogue only (splitk_finish and store_tile_rows share their arithmetic), and only forms
  // whose second accumulator set fits the register file (the register-staged 512-pixel form does not)
  P.vsplit = (force_splits > 0 && P.splits > 1 && own_splits == 1 && P.d_bstride == 0 && !P.noise &&
              !(CT_TILES == 2 && !P.xh)) ? 1 : 0;
  // the checks that can still reject this tile form come BEFORE the plan-only return: the workspace query must plan with
  // the form the launch will take
  dim3 grid(geom_blocks(G), P.co_tiles * groups, P.vsplit ? 1 : P.splits);
  if (grid.y > 65535) return HF_E_INVALID;
  // Block order (blocks are dispatched x-fastest, 256 at a time): tiles-fastest keeps a column's weights in L2 and re-reads every
  // input tile once per column from beyond L2 (columns x input bytes); columns-fastest keeps ~256 / columns input tiles in L2 and
  // re-reads the weights once per such group (weights fit the 256 MB Infinity Cache; the input of a batched pass does not).
  // Taken when it moves fewer bytes from beyond L2 and no counters are keyed by (x, y) (real split-K keeps tiles-fastest).
  P.swap_xy = 0;
  if (HF_ENC_SWAP_XY && (P.vsplit || P.splits == 1) && grid.x <= 65535 && grid.y > 1) {
    const double col_in = (double)P.batch * P.cin * P.h * P.w * 4.0;   // input bytes one column reads (hi + lo, or fp32)
    const double in_bytes = col_in * (P.x_gstride ? groups : 1);       // all inputs once
    const double w_bytes = 9.0 * P.cin * P.cout * 4.0 * groups;        // all weights once (hi + lo)
    const double tiles_fast = (double)grid.y * col_in + w_bytes;
    const double resident = grid.y >= 256 ? 1.0 : 256.0 / grid.y;      // input tiles in flight at a time
    const double cols_fast = in_bytes + w_bytes * ((double)grid.x / resident);
    // (an input that fits the 256 MB Infinity Cache beside the rest is re-read from there either way: from half of it upward;
    // hf_debug_set_tuning bit 3: tests force the order)
    if ((g_h_tune & 8) || (col_in > 128e6 && cols_fast < 0.5 * tiles_fast)) {
      P.swap_xy = 1;
      grid = dim3(grid.y, grid.x, grid.z);
    }
  }
  const size_t lds = (size_t)2 * NPART * (9 * 2 * CT + 2 * NPIX) * 16 + 2 * ((P.cin + 3) & ~3) * sizeof(float);
  if (lds > 160 * 1024) return HF_E_INVALID;
  // (every rejection of the tile form sits above the plan-only return: a query - hf_conv2d_f16_workspace_floats,
  // hf_conv2d_f16_split_output_ok, which plans with P.oh set - walks the same fall-through chain as the launch)
  if (P.oh && ((P.splits > 1 && !P.vsplit) || groups > 1)) return HF_E_INVALID;  // split output: written by this kernel's own epilogue only
  if (P.xh && ((long long)2 * P.h * P.w * 16 >= (1LL << 31) || (NTERMS == 3 && !P.xl))) return HF_E_INVALID;
  if (plan_only) return HF_OK;
  if (P.splits > 1 && !P.vsplit) {
    if (!workspace || workspace_floats < P.splits * P.zslab) return HF_E_WORKSPACE;
    P.partial = workspace;
    P.counters = splitk_counters_for((long long)grid.x * grid.y);
  }
  g_s2mt_last = 0;
  if (P.xh) {
    if constexpr (STRIDE == 2 && HF_ENC_S2MT > 1) {
      // several pixel tiles per resident weight stage (conv_enc_s2mt_h) when the launch still fills the chip with MT times
      // fewer blocks (hf_debug_set_tuning bits 24-31 lower "the chip" for tests) and its K loop is not spread over the grid
      // MT = 4, else 2: the largest one whose blocks still fill the chip's rounds of 256 resident blocks (512 @ 32^2 at batch 96:
      // 1536 one-tile blocks = 6 full rounds, 384 four-tile blocks = 1.5 - measured 17 % slower; 768 two-tile blocks = 3)
      const long long fill_blocks = fill ? fill : 256;
      const int cols = P.co_tiles * groups, tiles = geom_blocks(G);
      int mt = 0;
      if (P.vsplit || P.splits == 1) {
        for (int m = HF_ENC_S2MT; m >= 2 && !mt; m >>= 1) {
          const long long nb = (long long)hf_cdiv(tiles, m) * cols;
          const long long rounds = (nb + fill_blocks - 1) / fill_blocks;
          if (nb >= fill_blocks && (double)nb >= 0.85 * (double)(rounds * fill_blocks)) mt = m;
        }
      }
      if (mt) {
        const int tb = hf_cdiv(tiles, mt);
        dim3 g2 = P.swap_xy ? dim3(cols, tb, 1) : dim3(tb, cols, 1);
        if (mt == 4) {
          if (P.vsplit) hipLaunchKernelGGL((conv_enc_s2mt_h<NTERMS, 4, true>), g2, dim3(NT), lds, st, P, wth, wtl);
          else hipLaunchKernelGGL((conv_enc_s2mt_h<NTERMS, 4, false>), g2, dim3(NT), lds, st, P, wth, wtl);
        } else {
          if (P.vsplit) hipLaunchKernelGGL((conv_enc_s2mt_h<NTERMS, 2, true>), g2, dim3(NT), lds, st, P, wth, wtl);
          else hipLaunchKernelGGL((conv_enc_s2mt_h<NTERMS, 2, false>), g2, dim3(NT), lds, st, P, wth, wtl);
        }
        g_s2mt_last = mt;
        return hf_launch_status();
      }
    }
  }
  // persistent form: one resident block per CU walks the grid (forms whose LDS leaves room for one block per CU only)
  P.persist = 0;
  {
    static const int on = [] { const char *e = getenv("HAIRFAST_ENC_PERSIST"); return e ? atoi(e) : HF_ENC_PERSIST; }();
    const int resident = g_h_blocks > 0 ? g_h_blocks : 256;
    const bool ok = P.vsplit ? enc_persist_ok<NTERMS, CT_TILES, true>() : enc_persist_ok<NTERMS, CT_TILES, false>();
    if (on && ok && (P.vsplit || P.splits == 1) && lds > 80 * 1024 && blocks > resident) {
      P.persist = resident;
      P.lin_x = (int)grid.x;
      P.lin_y = (int)grid.y;
      grid = dim3(resident, 1, 1);
    }
  }
  if (P.xh) {
    if (P.vsplit)
      hipLaunchKernelGGL((conv_enc_h<NTERMS, PG, WAVES_PX, STRIDE, true, CT_TILES, WAVES_CO, true>), grid, dim3(NT), lds, st, P, wth, wtl);
    else
      hipLaunchKernelGGL((conv_enc_h<NTERMS, PG, WAVES_PX, STRIDE, true, CT_TILES, WAVES_CO>), grid, dim3(NT), lds, st, P, wth, wtl);
  } else if (P.vsplit) {
    if constexpr (CT_TILES == 2) return HF_E_INVALID;  // (never planned: see P.vsplit above)
    else
      hipLaunchKernelGGL((conv_enc_h<NTERMS, PG, WAVES_PX, STRIDE, false, CT_TILES, WAVES_CO, true>), grid, dim3(NT), lds, st, P, wth, wtl);
  } else {
    hipLaunchKernelGGL((conv_enc_h<NTERMS, PG, WAVES_PX, STRIDE, false, CT_TILES, WAVES_CO>), grid, dim3(NT), lds, st, P, wth, wtl);
  }
  int rc = hf_launch_status();
  if (rc == HF_OK && P.splits > 1 && !P.vsplit && !P.counters) rc = launch_splitk_reduce(P, true, st);  // deterministic second pass + epilogue
  return rc;
}

// the tile configuration hf_conv2d_f16_f32 uses for a shape (shared by the workspace query).  Tile forms follow the real
// launch in every mode: they do not change the K order (equal bits, tests/test_sim_encoders.py).
template <int NTERMS>
int run_enc_forms(ConvParams &P, const _Float16 *hi, const _Float16 *lo, float *ws, long long wsn, hipStream_t st, bool plan_only,
                  int force_splits) {
  int rc;
  if (P.stride == 2) {
    // 64 co x 128 output px (the halo of a stride-2 tile is 4x its output: a 256-pixel tile does not fit twice in LDS).
    // Eight waves of 1 x 1 MFMA tiles: two waves per SIMD hide the stage latencies, which is worth more here than the
    // 1 instead of 1.33 LDS fragment reads per MFMA of four 1 x 2 waves (tools/probes/stride2.py: 10-25% faster from
    // 64@256^2 to the 11-group style heads, 30-35% on the register-staged path; same accumulation order, equal bits)
    rc = launch_enc<NTERMS, 1, 4, 2>(P, hi, lo, ws, wsn, st, plan_only, force_splits);
    if (rc == HF_OK && !plan_only) note_path(6, g_s2mt_last == 4 ? 5 : g_s2mt_last ? 6 : 2);
    return rc;
  }
  // 64 co x 512 px (8 waves, 2 x 2 MFMA tiles each) when that fills the chip (batched swaps), else
  // 64 co x 256 px (8 waves, 1 x 2 tiles) when that still does, else 64 co x 128 px (4 waves)
  const int groups = P.groups > 1 ? P.groups : 1;
  const long long blocks256 = (long long)P.batch * groups * hf_cdiv((long long)P.out_h * P.out_w, 256) * (P.cout / 64);
  const long long blocks512 = (long long)P.batch * groups * hf_cdiv((long long)P.out_h * P.out_w, 512) * (P.cout / 64);
  rc = HF_E_INVALID;
  // hf_debug_set_tuning: bit 2 = never the 512-pixel form, bits 8-15 = its minimum block count / 8 (0: the default),
  // bits 16-23 = the minimum block count / 8 of the 256-pixel form
  const int min512 = ((g_h_tune >> 8) & 255) > 0 ? ((g_h_tune >> 8) & 255) * 8 : 512;
  const int min256 = ((g_h_tune >> 16) & 255) > 0 ? ((g_h_tune >> 16) & 255) * 8 : 384;
  // (a given K partition of more than one slab - batch-invariant plans - on the register-staged 512-pixel form cannot run
  // virtually: it would spread force_splits slabs of the whole batched output over the grid plus a reduce pass; the 256-pixel
  // form below walks them inside its blocks - same bits)
  if (NTERMS == 3 && !(g_h_tune & 4) && blocks512 >= min512 && !(force_splits > 1 && !P.xh)) {  // plain fp16 operands: staging-bound, measured slower
    rc = launch_enc<NTERMS, 2, 8, 1, 2, 1>(P, hi, lo, ws, wsn, st, plan_only, force_splits);
    if (rc == HF_OK && !plan_only) note_path(6, 4);
  }
  if (rc == HF_E_INVALID && blocks256 >= min256) {
    rc = launch_enc<NTERMS, 2, 4, 1>(P, hi, lo, ws, wsn, st, plan_only, force_splits);
    if (rc == HF_OK && !plan_only) note_path(6, 1);
  }
  if (rc == HF_E_INVALID) {
    rc = launch_enc<NTERMS, 2, 2, 1>(P, hi, lo, ws, wsn, st, plan_only, force_splits);
    if (rc == HF_OK && !plan_only) note_path(6, 3);
  }
  return rc;
}

// Batch-invariant mode: the K partition - what decides a sample's bits - is the one the CANONICAL launch of this layer
// (batch kCanonBatch, whatever the real batch) plans for itself; the real launch then runs that partition in its own tile form.
template <int NTERMS>
int run_enc(ConvParams &P, const _Float16 *hi, const _Float16 *lo, float *ws, long long wsn, hipStream_t st, bool plan_only) {
  int force = 0;
  if (g_batch_invariant) {
    ConvParams C = P;
    C.batch = kCanonBatch;
    C.oh = C.ol = nullptr;  // only the canonical K partition is wanted: whether THIS launch can write a split output is decided below
    const int rc = run_enc_forms<NTERMS>(C, hi, lo, nullptr, 0, st, true, 0);
    if (rc != HF_OK) return rc;
    force = C.splits;
  }
  return run_enc_forms<NTERMS>(P, hi, lo, ws, wsn, st, plan_only, force);
}

// s*x + t (per input channel; null = identity) split into fp16 (hi, lo) and K-blocked
// [image][c/8][h][w][8 halves]: one thread per (image, channel block, pixel), 8 coalesced plane loads
__global__ __launch_bounds__(256) void split_activation(half8 *__restrict__ hi, half8 *__restrict__ lo,
                                                        const float *__restrict__ x, const float *__restrict__ s,
                                                        const float *__restrict__ t, int channels, long long plane,
                                                        long long total, int s_istride) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  const int cblocks = channels >> 3;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long long pix = i % plane, r = i / plane;
    const int cb = (int)(r % cblocks);
    const long long im = r / cblocks;
    const float *src = x + (im * channels + cb * 8) * plane + pix;
    half8 h8, l8;
    bool ovf = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = cb * 8 + k;
      const float v = fmaf(src[k * plane], s ? s[im * s_istride + c] : 1.0f, t ? t[c] : 0.0f);
      _Float16 hv, lv;
      hf_split_f16(v, hv, lv, ovf);
      h8[k] = hv;
      l8[k] = lv;
    }
    hf_note_overflow(ovf);
    hi[i] = h8;
    if (lo) lo[i] = l8;
  }
}

// The tail of a bottleneck_IR_SE unit (helpers.py:118-120: res * gate + shortcut, shortcut optionally MaxPool2d(1, stride)) that
// ALSO leaves as the pre-split input of the next unit's first conv: out = r * gate + shortcut (fp32, the next unit's shortcut), and
// s*out + t (the next unit's BatchNorm) split into fp16 (hi, lo), K-blocked - bit for bit what hf_scale_shortcut_add_f32 followed by
// hf_split_activation_f16 write.  One thread per (image, channel block of 8, pixel): 8 coalesced plane loads of r and of the shortcut.
__global__ __launch_bounds__(256) void scale_shortcut_add_split(float *__restrict__ out, half8 *__restrict__ hi, half8 *__restrict__ lo,
                                                                const float *__restrict__ r, const float *__restrict__ gate,
                                                                const float *__restrict__ shortcut, const float *__restrict__ s,
                                                                const float *__restrict__ t, int sc_stride, int channels, int oh,
                                                                int ow, int sh, int sw, long long total) {
#pragma clang fp contract(on)
  const long long stride = (long long)gridDim.x * blockDim.x;
  const int cblocks = channels >> 3, oplane = oh * ow;
  const long long splane = (long long)sh * sw;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int pix = (int)(i % oplane);
    const long long q = i / oplane;
    const int cb = (int)(q % cblocks);
    const long long im = q / cblocks;
    const int y = pix / ow, x = pix - y * ow;
    const long long pl0 = im * channels + cb * 8;
    const float *rp = r + pl0 * oplane + pix;
    const float *sp = shortcut + pl0 * splane + (long long)(y * sc_stride) * sw + x * sc_stride;
    float *op = out + pl0 * oplane + pix;
    half8 h8, l8;
    bool ovf = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = cb * 8 + k;
      float v = rp[(long long)k * oplane];
      if (gate) v *= gate[pl0 + k];
      v += sp[(long long)k * splane];
      op[(long long)k * oplane] = v;
      const float u = fmaf(v, s ? s[c] : 1.0f, t ? t[c] : 0.0f);
      _Float16 hv, lv;
      hf_split_f16(u, hv, lv, ovf);
      h8[k] = hv;
      l8[k] = lv;
    }
    hf_note_overflow(ovf);
    hi[i] = h8;
    if (lo) lo[i] = l8;
  }
}

}  // namespace

extern "C" int hf_scale_shortcut_add_split_f16(float *out, void *out_hi, void *out_lo, const float *next_scale,
                                               const float *next_shift, const float *r, const float *gate, const float *shortcut,
                                               int sc_stride, int batch, int channels, int oh, int ow, int sh, int sw, void *stream) {
  if (!out || !out_hi || !r || !shortcut || sc_stride < 1 || batch <= 0 || channels <= 0 || (channels & 7) || oh <= 0 || ow <= 0)
    return HF_E_INVALID;
  if ((oh - 1) * sc_stride >= sh || (ow - 1) * sc_stride >= sw) return HF_E_INVALID;
  const long long total = (long long)batch * (channels >> 3) * oh * ow;
  long long g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(scale_shortcut_add_split, dim3((int)g), dim3(256), 0, (hipStream_t)stream, out, static_cast<half8 *>(out_hi),
                     static_cast<half8 *>(out_lo), r, gate, shortcut, next_scale, next_shift, sc_stride, channels, oh, ow, sh, sw, total);
  return hf_launch_status();
}

extern "C" unsigned long long hf_f16_overflow_count_enc(int reset) { return hf_f16_overflow_read_tu(reset); }

static int enc_fill(ConvParams &P, float *out, const float *x, const void *x_hi, const void *x_lo, const float *in_scale,
                    const float *in_shift, const float *out_scale, const float *bias, int act, const float *slope,
                    float alpha, const float *residual, int batch, int cin, int cout, int h, int w, int stride, int groups,
                    long long x_group_stride) {
  if (batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0 || (stride != 1 && stride != 2) || groups < 1 || x_group_stride < 0)
    return HF_E_INVALID;
  P.residual_pre = (act & HF_ACT_RESIDUAL_FIRST) ? 1 : 0;
  act &= ~HF_ACT_RESIDUAL_FIRST;
  if (act < ACT_NONE || act > ACT_QGELU || (act == ACT_PRELU && !slope)) return HF_E_INVALID;
  if (groups > 1 && (in_scale || in_shift)) return HF_E_INVALID;  // grouped form: plain conv + epilogue (as hf_conv2d_f32)
  if (x_hi && (in_scale || in_shift)) return HF_E_INVALID;        // pre-split input: the affine went into the split
  P.out = out; P.x = x; P.xh = x_hi; P.xl = x_lo; P.s = in_scale; P.t = in_shift; P.d = out_scale; P.bias = bias;
  P.slope = slope; P.residual = residual;
  P.s_bstride = 0; P.d_bstride = 0;
  P.batch = batch; P.cin = cin; P.cout = cout; P.h = h; P.w = w;
  P.out_h = (h - 1) / stride + 1; P.out_w = (w - 1) / stride + 1; P.out_wv = P.out_w;
  P.stride = stride;
  P.act = act; P.alpha = alpha; P.scale = 1.0f;
  P.groups = groups; P.x_gstride = x_group_stride; P.wt_gstride = 0;
  return HF_OK;
}

extern "C" long long hf_conv2d_f16_workspace_floats(int batch, int cin, int cout, int h, int w, int stride, int groups) {
  ConvParams P{};
  if (enc_fill(P, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, ACT_NONE, nullptr, 0.0f, nullptr, batch,
               cin, cout, h, w, stride, groups, 0) != HF_OK)
    return 0;
  // both operand modes (their tile forms can differ: nterms 1 never takes the 512-pixel form): the larger plan
  long long need = 0;
  // (planned for register-staged input: a pre-split launch may split virtually where this one splits for real - an upper bound)
  if (run_enc<3>(P, nullptr, nullptr, nullptr, 0, nullptr, true) == HF_OK && P.splits > 1 && !P.vsplit) need = P.splits * P.zslab;
  ConvParams Q = P;
  if (run_enc<1>(Q, nullptr, nullptr, nullptr, 0, nullptr, true) == HF_OK && Q.splits > 1 && !Q.vsplit && Q.splits * Q.zslab > need)
    need = Q.splits * Q.zslab;
  return need;
}

// 1 when hf_conv2d_f16_split_f32 takes the launch (its epilogue must run in the conv kernel itself: no real split-K; a
// virtual one - batch-invariant plans, ConvParams::vsplit - is fine, and whether a launch can split virtually depends on
// how its input arrives), else 0.
extern "C" int hf_conv2d_f16_split_output_ok(int batch, int cin, int cout, int h, int w, int stride, int nterms, int presplit_input) {
  ConvParams P{};
  if ((nterms != 1 && nterms != 3) || (cout & 7) ||
      enc_fill(P, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, ACT_NONE, nullptr, 0.0f, nullptr, batch,
               cin, cout, h, w, stride, 1, 0) != HF_OK)
    return 0;
  if (presplit_input) P.xh = P.xl = &P;  // only tested for NULL while planning
  P.oh = &P;                             // a split output is wanted: forms that cannot write one are skipped like in the launch
  const int rc = (nterms == 3) ? run_enc<3>(P, nullptr, nullptr, nullptr, 0, nullptr, true) : run_enc<1>(P, nullptr, nullptr, nullptr, 0, nullptr, true);
  return (rc == HF_OK && !(P.splits > 1 && !P.vsplit)) ? 1 : 0;
}

extern "C" int hf_conv2d_f16_f32(float *out, const float *x, const void *x_hi, const void *x_lo, const void *wt_hi,
                                 const void *wt_lo, int nterms, const float *in_scale, const float *in_shift,
                                 const float *out_scale, const float *bias, int act, const float *slope, float alpha,
                                 const float *residual, int batch, int cin, int cout, int h, int w, int stride, int groups,
                                 long long x_group_stride, float *workspace, long long workspace_floats, void *stream) {
  if (!out || (!x && !x_hi) || !wt_hi || (nterms != 1 && nterms != 3) || (nterms == 3 && !wt_lo)) return HF_E_INVALID;
  ConvParams P{};
  int rc = enc_fill(P, out, x, x_hi, x_lo, in_scale, in_shift, out_scale, bias, act, slope, alpha, residual, batch, cin, cout, h,
                    w, stride, groups, x_group_stride);
  if (rc != HF_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  const _Float16 *hi = static_cast<const _Float16 *>(wt_hi), *lo = static_cast<const _Float16 *>(wt_lo);
  return (nterms == 3) ? run_enc<3>(P, hi, lo, workspace, workspace_floats, st, false)
                       : run_enc<1>(P, hi, lo, workspace, workspace_floats, st, false);
}

// hf_conv2d_f16_f32 whose result ALSO (or only: out NULL) leaves as the pre-split input of the next fp16-core conv:
// next_scale[co] * y + next_shift[co] as fp16 (hi, lo), K-blocked - what hf_split_activation_f16 would make of y in a
// pass of its own.  Launches that would run split-K (hf_conv2d_f16_workspace_floats != 0) or groups are refused.
extern "C" int hf_conv2d_f16_split_f32(float *out, void *out_hi, void *out_lo, const float *next_scale, const float *next_shift,
                                       const float *x, const void *x_hi, const void *x_lo, const void *wt_hi, const void *wt_lo,
                                       int nterms, const float *in_scale, const float *in_shift, const float *out_scale,
                                       const float *bias, int act, const float *slope, float alpha, const float *residual,
                                       int batch, int cin, int cout, int h, int w, int stride, void *stream) {
  if (!out_hi || (!x && !x_hi) || !wt_hi || (nterms != 1 && nterms != 3) || (nterms == 3 && (!wt_lo || !out_lo)) || (cout & 7))
    return HF_E_INVALID;
  ConvParams P{};
  int rc = enc_fill(P, out, x, x_hi, x_lo, in_scale, in_shift, out_scale, bias, act, slope, alpha, residual, batch, cin, cout, h,
                    w, stride, 1, 0);
  if (rc != HF_OK) return rc;
  P.oh = out_hi; P.ol = out_lo; P.a_next = next_scale; P.t_next = next_shift;
  if ((long long)batch * (cout >> 3) * P.out_h * P.out_w >= (1LL << 40)) return HF_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const _Float16 *hi = static_cast<const _Float16 *>(wt_hi), *lo = static_cast<const _Float16 *>(wt_lo);
  return (nterms == 3) ? run_enc<3>(P, hi, lo, nullptr, 0, st, false) : run_enc<1>(P, hi, lo, nullptr, 0, st, false);
}

extern "C" int hf_split_activation_f16(void *out_hi, void *out_lo, const float *x, const float *in_scale, const float *in_shift,
                                       long long images, int channels, int h, int w, void *stream) {
  if (!out_hi || !x || images <= 0 || channels <= 0 || (channels & 7) || h <= 0 || w <= 0) return HF_E_INVALID;
  const long long plane = (long long)h * w, total = images * (channels >> 3) * plane;
  long long g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(split_activation, dim3((int)g), dim3(256), 0, (hipStream_t)stream, static_cast<half8 *>(out_hi),
                     static_cast<half8 *>(out_lo), x, in_scale, in_shift, channels, plane, total, 0);
  return hf_launch_status();
}

// hf_split_activation_f16 with a PER-IMAGE scale s[image][channel] (the modulation of a ModulatedConv2d applied to its
// input, models/stylegan2/model.py:241-248) instead of the per-channel affine
extern "C" int hf_split_activation_mod_f16(void *out_hi, void *out_lo, const float *x, const float *scale, long long images,
                                           int channels, int h, int w, void *stream) {
  if (!out_hi || !x || images <= 0 || channels <= 0 || (channels & 7) || h <= 0 || w <= 0) return HF_E_INVALID;
  const long long plane = (long long)h * w, total = images * (channels >> 3) * plane;
  long long g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(split_activation, dim3((int)g), dim3(256), 0, (hipStream_t)stream, static_cast<half8 *>(out_hi),
                     static_cast<half8 *>(out_lo), x, scale, (const float *)nullptr, channels, plane, total, scale ? channels : 0);
  return hf_launch_status();
}

#ifdef HF_ENC_TRACE
extern "C" int hf_debug_read_enc_trace(unsigned long long *host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(hf_enc_trace_buf), sizeof(unsigned long long) * n);
}
#endif
