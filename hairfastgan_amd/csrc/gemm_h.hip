// gemm_h.hip - 1x1 convolutions / Linear layers as GEMMs on the fp16 matrix cores (v_mfma_f32_32x32x16_f16).
//
// Same arithmetic contract as hf_conv2d_f32 with k = 1 (csrc/modconv.hip):
//   y[b,co,p] = act( out_scale[co] * sum_ci W[co,ci] * (in_scale[ci]*x[b,ci,p'] + in_shift[ci]) + bias[co] ) + residual
// (p' = the stride-1 / stride-2 source pixel of output pixel p) in the operand modes of csrc/convh.hip: nterms 3 = fp32
// operands split into fp16 (hi, lo) pairs, hi*hi + hi*lo + lo*hi in the fp32 accumulator (fp32-class accuracy), nterms 1
// = operands rounded to fp16.  Reference operators: the 1x1 shortcut / downsample convolutions of the encoders
// (models/encoder4editing/models/encoders/helpers.py:99-103, models/FeatureStyleEncoder/arcface/iresnet.py:17-19,
// BiSeNet's 1x1 convs, model.py:13-29), SEAN's conv_s (networks/architecture.py:50, 88-92), and - on feature-major
// activations x[feature][token], an NCHW tensor whose pixels are the tokens - every nn.Linear of the CLIP ViT-B/32 image
// tower (clip/model.py: in_proj / out_proj / c_fc / c_proj / the patch embedding / proj) and SEAN's per-label table GEMM.
//
// Structure: one block = 64 output channels x PT = 64*PG pixels (of one image, or of 2^n whole small images), 4 waves =
// 2 (channels) x 2 (pixels), each 1 x PG MFMA tiles; K loop in stages of 32 input channels, double-buffered in LDS:
// weights by LDS-DMA from [cin/16][kgroup 2][cout][8 halves] hi (+ lo) (hf_conv_split_weights_f16 with taps = 1,
// pre-scaled by 2^k, un-scaled in the epilogue), activations through registers (8 coalesced plane loads per
// (pixel, 8-channel block) item, affine, saturating split, one 16-byte LDS write per part) loaded one stage ahead.
// Small grids split K over blockIdx.z (deterministic second pass: splitk_reduce of modconv.hip).
#define HF_WANT_F16_SPLIT
#include "conv_common.h"

using namespace hf_detail;

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
constexpr int KH = 16;   // input channels per MFMA k-step
constexpr int KS = 32;   // input channels per LDS stage
// Two blocks per CU: the 64 x 256 form's two stage buffers are exactly 80 KiB - half of a CU's 160 KiB - but without a register bound
// hipcc spends 290 registers on it (one wave per SIMD, every LDS / DMA latency exposed); bounded to two waves per SIMD it needs
// 226-240 and no scratch (round 5; -DHF_GEMM_MIN_WAVES=1 = the old build for A/B).
#ifndef HF_GEMM_MIN_WAVES
#define HF_GEMM_MIN_WAVES 2
#endif
// blocks from which a launch fills the chip without a K split (hf_debug_set_tuning bits 24-31 lower it for tests)
inline int gemm_fill_blocks() { return ((hf_detail::g_h_tune >> 24) & 255) ? ((hf_detail::g_h_tune >> 24) & 255) : 256; }

// PRE: the activations arrive already transformed, split into fp16 (hi, lo) and K-blocked ([image][cin/8][h*w][8 halves]:
// hf_split_activation_f16 / hf_split_activation_mod_f16; ConvParams::xh / xl) and are staged by LDS-DMA like the weights -
// an input shared by many output-channel tiles (the tap-GEMM's 72, the CLIP MLP's 48) is converted once, not once per block.
// VSPLIT (ConvParams::vsplit): the block walks all P.splits K slabs and adds their sums in z order to a second accumulator
// set - the bits of the split-K form (slabs + splitk_reduce / small_combine) without the slabs (see csrc/convh_enc.hip).
// CW (round 6): output-channel waves of a block - 2: 64 channels x PT pixels, four waves, two blocks per CU; 4: 128 channels x PT
// pixels, eight waves, one block per CU (96 KB): the activation stage (32 KB of the 40 a 64-channel block copies per 96 MFMAs) is
// shared by twice the channels - 48 KB per 192 MFMAs, 250 instead of 427 bytes of LDS-DMA per MFMA.  Same K order: equal bits.
template <int NTERMS, int PG, bool PRE, bool VSPLIT = false, int CW = 2>
__global__ __launch_bounds__(128 * CW, HF_GEMM_MIN_WAVES) void gemm1x1_h(const ConvParams P, const _Float16 *__restrict__ wth_all,
                                                 const _Float16 *__restrict__ wtl_all) {
  constexpr int NW = 2 * CW, NT = 64 * NW, CT = 32 * CW, PT = 64 * PG;
  constexpr int NPART = (NTERMS == 3) ? 2 : 1;
  constexpr int W_UNITS = (KS / 8) * CT;        // 16-byte units of one weight part per stage: [chunk 2][kg 2][64 co]
  constexpr int X_UNITS = (KS / 8) * PT;        // [kblock 4][PT pixels]
  constexpr int BUF_UNITS = NPART * (W_UNITS + X_UNITS);
  constexpr int OFF_WL = W_UNITS, OFF_XH = NPART * W_UNITS, OFF_XL = NPART * W_UNITS + X_UNITS;
  constexpr int XE = X_UNITS / NT;              // staging items per thread and stage (PG, or PG / 2 with eight waves)
  constexpr int N_WPIECE = NPART * W_UNITS / 64;  // 1 KiB DMA pieces per stage (4 or 8; 8 or 16 with 128 channels)
  constexpr int ND = N_WPIECE / NW;             // per wave
  static_assert(X_UNITS % NT == 0 && N_WPIECE % NW == 0 && XE >= 1, "whole staging items / DMA pieces per thread / wave");

  HF_DYN_LDS;
  half8 *lds = reinterpret_cast<half8 *>(hf_dyn_lds);  // [2][BUF_UNITS]
  const GroupOfs go = group_offsets(P);
  const int grp = (P.groups > 1) ? (int)blockIdx.y / P.co_tiles : 0;
  const long long wn = (long long)P.cin * P.cout;
  const _Float16 *wth = wth_all + (long long)grp * (wn + 8);  // [weights | 16-byte trailer] per group
  const _Float16 *wtl = wtl_all ? wtl_all + (long long)grp * wn : nullptr;
  const float w_unscale = *reinterpret_cast<const float *>(wth + wn);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wave_co = (wave >> 1) * 32, wave_pg = (wave & 1) * PG;
  const int co0 = go.co_tile * CT;

  // ---- tile: PT flat OUTPUT pixels (the epilogue sees every image as a 1 x (oh*ow) plane) ----
  const TileGeom G = P.g[0];
  const int oplane = P.out_h * P.out_w;  // = G.dw
  int t = blockIdx.x;
  const int tx = t % G.tiles_x;
  const int b0 = (t / G.tiles_x) << G.lg_nb;
  const int tx0 = tx << G.lg_tw;
  const long long iplane = (long long)P.h * P.w;
  const float *xb = PRE ? nullptr : P.x + go.x;

  // staging items (stage invariant): element offset of the source pixel inside image 0's channel 0 plane (+ image
  // offset), -1 = outside (zero)
  long long e_src[XE];
  long long e_unit[XE];  // PRE: 16-byte unit of the pixel in channel block 0 of its image
  int e_sofs[XE];  // offset of the item's image inside in_scale (s_bstride != 0: a per-image scale, the modulation)
#pragma unroll
  for (int e = 0; e < XE; ++e) {
    const int i = tid + e * NT;
    const int px = i % PT;  // i / PT = kblock
    const int pp = px & ((1 << G.lg_tw) - 1), im = px >> G.lg_tw;
    const int p = tx0 + pp, b = b0 + im;
    e_src[e] = -1;
    e_unit[e] = 0;
    e_sofs[e] = (b < P.batch ? b : 0) * P.s_bstride;
    if (p < oplane && b < P.batch) {
      const int oy = p / P.out_wv, ox = p - oy * P.out_wv;  // out_wv: the true output width (out_w is the flat plane)
      e_src[e] = (long long)b * P.cin * iplane + (long long)(oy * P.stride) * P.w + ox * P.stride;
      e_unit[e] = (long long)b * (P.cin >> 3) * iplane + p;
    }
  }

  const int nstages_all = P.cin / KS;
  const int s_begin = (P.splits > 1 && !VSPLIT) ? (int)blockIdx.z * P.chunks_per_split : 0;
  const int s_end = (P.splits > 1 && !VSPLIT) ? min(nstages_all, s_begin + P.chunks_per_split) : nstages_all;

  const unsigned lds_addr0 = hf_lds_addr(lds);
  auto dma_w = [&](int stage, int bufsel) {
#pragma unroll
    for (int j = 0; j < ND; ++j) {
      const int pc = wave + j * NW;  // piece: (part, chunk-in-stage, kg[, 64-channel half])
      const int part = pc / (W_UNITS / 64), q = pc % (W_UNITS / 64);  // q = (chunk*2 + kg) * (CT / 64) + 64-channel half
      const int row = q / (CT / 64), half = q % (CT / 64);
      const _Float16 *src = (part ? wtl : wth) + ((long long)(stage * (KS / KH) * 2 + row) * P.cout + co0 + half * 64) * 8;
      hf_glds16_raw_s(src, (unsigned)lane * 16u, lds_addr0 + (unsigned)(bufsel * BUF_UNITS + part * W_UNITS + q * 64) * 16u);
    }
  };
  // PRE: one 16-byte unit per lane and item straight into LDS (stride 1: e_src is b*cin*plane + p -> unit (b*cin/8 + kblock)*plane + p);
  // items outside the image are masked out of the DMA and stay zero (filled once below)
  auto dma_x = [&](int stage, int bufsel) {
#pragma unroll
    for (int e = 0; e < XE; ++e) {
      const int i = tid + e * NT;
      const int kb = i / PT;
      const bool inside = e_src[e] >= 0;
      const long long unit = e_unit[e] + (long long)(stage * (KS / 8) + kb) * iplane;
      const unsigned off = (unsigned)(unit * 16);
      const unsigned dst = lds_addr0 + (unsigned)(bufsel * BUF_UNITS + OFF_XH + (i - lane)) * 16u;
      hf_glds16_raw_s_if(inside, P.xh, off, dst);
      if (NTERMS == 3) hf_glds16_raw_s_if(inside, P.xl, off, dst + (unsigned)X_UNITS * 16u);
    }
  };
  float xr[PRE ? 1 : XE][8];
  auto load_x = [&](int stage) {
    if (PRE) return;
#pragma unroll
    for (int e = 0; e < XE; ++e) {
      const int kb = (tid + e * NT) / PT;
      const float *src = xb + (e_src[e] >= 0 ? e_src[e] : 0) + (long long)(stage * KS + kb * 8) * iplane;
#pragma unroll
      for (int k = 0; k < 8; ++k) xr[PRE ? 0 : e][k] = src[(long long)k * iplane];
    }
  };
  auto convert_x = [&](int stage, half8 *buf) {
    if (PRE) return;
    bool ovf = false;
#pragma unroll
    for (int e = 0; e < XE; ++e) {
      const int i = tid + e * NT;
      const int kb = i / PT;
      half8 hi, lo;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int ci = stage * KS + kb * 8 + k;
        float v = xr[PRE ? 0 : e][k];
        v = fmaf(v, P.s ? P.s[e_sofs[e] + ci] : 1.0f, P.t ? P.t[ci] : 0.0f);  // as hf_split_activation_f16 rounds it
        if (e_src[e] < 0) v = 0.0f;
        _Float16 hv, lv;
        hf_split_f16(v, hv, lv, ovf);
        hi[k] = hv;
        lo[k] = lv;
      }
      buf[OFF_XH + i] = hi;
      if (NTERMS == 3) buf[OFF_XL + i] = lo;
    }
    hf_note_overflow(ovf);
  };

  f32x16 acc[1][1][PG];
  f32x16 vsum[VSPLIT ? PG : 1];        // VSPLIT: the slabs added so far
  int slab_left = P.chunks_per_split;  // VSPLIT: stages until the current slab ends
#pragma unroll
  for (int g = 0; g < PG; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc[0][0][g][r] = 0.0f;
      if (VSPLIT) vsum[VSPLIT ? g : 0][r] = 0.0f;
    }

  if (PRE) {  // items outside the image: zero in both buffers, once (the masked DMA never touches them)
    half8 z;
#pragma unroll
    for (int k = 0; k < 8; ++k) z[k] = (_Float16)0.0f;
#pragma unroll
    for (int e = 0; e < XE; ++e)
      if (e_src[e] < 0) {
        const int i = tid + e * NT;
#pragma unroll
        for (int bsel = 0; bsel < 2; ++bsel) {
          lds[bsel * BUF_UNITS + OFF_XH + i] = z;
          if (NTERMS == 3) lds[bsel * BUF_UNITS + OFF_XL + i] = z;
        }
      }
  }
  if (s_begin < s_end) {
    dma_w(s_begin, 0);
    if (PRE) dma_x(s_begin, 0);
    load_x(s_begin);
    convert_x(s_begin, lds);
  }
  hf_barrier_keep_young<0>();
  for (int s = s_begin; s < s_end; ++s) {
    const int cb = (s - s_begin) & 1;
    half8 *buf = lds + cb * BUF_UNITS, *nbuf = lds + (cb ^ 1) * BUF_UNITS;
    const bool more = s + 1 < s_end;
    if (more) {
      dma_w(s + 1, cb ^ 1);
      if (PRE) dma_x(s + 1, cb ^ 1);
      load_x(s + 1);
    }
#pragma unroll
    for (int cs = 0; cs < KS / KH; ++cs) {
      const half8 ah = buf[(cs * 2 + lh) * CT + wave_co + li];
      half8 al;
      if (NTERMS == 3) al = buf[OFF_WL + (cs * 2 + lh) * CT + wave_co + li];
#pragma unroll
      for (int g = 0; g < PG; ++g) {
        const int u = (cs * 2 + lh) * PT + (wave_pg + g) * 32 + li;
        const half8 bh = buf[OFF_XH + u];
        acc[0][0][g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[0][0][g], 0, 0, 0);
        if (NTERMS == 3) {
          const half8 bl = buf[OFF_XL + u];
          acc[0][0][g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[0][0][g], 0, 0, 0);
          acc[0][0][g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[0][0][g], 0, 0, 0);
        }
      }
    }
    if (more) convert_x(s + 1, nbuf);
    if (VSPLIT && (--slab_left == 0 || !more)) {  // a slab ends: its sum (pre-scale undone, exact) joins the earlier slabs'
      slab_left = P.chunks_per_split;
#pragma unroll
      for (int g = 0; g < PG; ++g) {
        vsum[VSPLIT ? g : 0] += acc[0][0][g] * w_unscale;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][g][r] = 0.0f;
      }
    }
    hf_barrier_keep_young<0>();
  }
  // epilogue: the weights' power-of-two pre-scale comes out here (exact); split-K launches store raw sums * 2^-k
#pragma unroll
  for (int g = 0; g < PG; ++g) {
    if (VSPLIT) acc[0][0][g] = vsum[VSPLIT ? g : 0];
    else
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][0][g][r] *= w_unscale;
  }
  store_tile_rows<1, PG>(P, G, go, acc, co0 + wave_co, wave_pg, li, lh, 0, tx0, b0);
  // split-K without a second launch: the tile's last block adds the slabs and runs the epilogue (conv_common.h)
  if (!VSPLIT && P.splits > 1 && P.counters && splitk_arrive_last(P, reinterpret_cast<int *>(hf_dyn_lds)))
    reduce_tile_rows<1, PG>(P, G, go, co0 + wave_co, wave_pg, li, lh, 0, tx0, b0);
}

thread_local int g_gemm_wide_last = 0;  // the last launch took the 128-channel block form (hf_debug_last_path 706)

template <int NTERMS, int PG>
int launch_gemm(ConvParams &P, const _Float16 *wth, const _Float16 *wtl, hipStream_t st) {
  constexpr int PT = 64 * PG;
  const int oplane = P.out_h * P.out_w;
  TileGeom g{};
  g.y0 = 0; g.x0 = 0; g.dh = 1; g.dw = oplane;
  int tw = PT, nb = 1;
  if (oplane < PT && (oplane & (oplane - 1)) == 0) {  // small power-of-two planes: several whole images per tile
    tw = oplane;
    nb = PT / oplane;
    if (nb > pow2_ceil(P.batch)) nb = pow2_ceil(P.batch);
  }
  g.lg_tw = ilog2(tw); g.lg_th = 0; g.lg_nb = ilog2(nb);
  g.tiles_x = hf_cdiv(oplane, tw); g.tiles_y = 1; g.tiles_b = hf_cdiv(P.batch, nb);
  g.first_block = 0;
  P.g[0] = g;
  P.n_geom = 1;
  const int nblocks = g.tiles_x * g.tiles_b;
  constexpr int NPART = (NTERMS == 3) ? 2 : 1;
  // 128-channel blocks (CW = 4) when the launch still fills the chip with them: a tile FORM - the K order of an output element
  // does not change - so it follows the real launch in every mode (tests/test_sim_gemm.py).  PG 4 only (the 256-pixel tile of
  // the large launches: with the 128-pixel tile - the heads' patch GEMM, 4608 -> 512 x 11 groups - it measured SLOWER, 2075 vs
  // 1840 us: four 48 KB blocks per CU hide more than two 64 KB ones); hf_debug_set_tuning bit 1 = never (A/B, tests)
  const int groups_ = max(1, P.groups);
  const bool wide = PG == 4 && !(g_h_tune & 2) && (P.cout % 128) == 0 &&
                    (long long)nblocks * (P.cout / 128) * groups_ >= 2LL * gemm_fill_blocks();  // (from ONE round of CUs: measured mixed - CLIP fc 66-70 -> 71-75 us, proj 101 -> 92-95, r06ab)
  if (wide) P.co_tiles = P.cout / 128;
  g_gemm_wide_last = wide ? 1 : 0;
  const int ct = wide ? 128 : 64;
  const size_t lds = (size_t)2 * NPART * ((KS / 8) * ct + (KS / 8) * PT) * 16;
  // batch-invariant plans: P.splits is the canonical K partition (gemm_splits at kCanonBatch); a launch whose output grid
  // fills the chip by itself walks the slabs inside its blocks instead of spreading them over grid.z (same bits)
  P.vsplit = (g_batch_invariant && P.splits > 1 && (long long)nblocks * P.co_tiles * max(1, P.groups) >= gemm_fill_blocks()) ? 1 : 0;
  dim3 grid(nblocks, P.co_tiles * max(1, P.groups), P.vsplit ? 1 : P.splits);
  if (grid.y > 65535 || grid.z > 65535) return HF_E_INVALID;
  // split-K of hf_conv1x1_f16_f32 (P.out = the result tensor): the in-kernel second half when a counter buffer is registered;
  // the tap-GEMM of the small-plane convs (P.out = its own slab workspace, combined by small_combine) keeps raw slabs
  P.counters = (P.splits > 1 && !P.vsplit && P.out != P.partial) ? splitk_counters_for((long long)grid.x * grid.y) : nullptr;
  if (P.xh) {
    if (P.stride != 1 || P.s || P.t || (NTERMS == 3 && !P.xl) || P.groups > 1 ||
        (long long)P.batch * P.cin * P.h * P.w * 2 >= (1LL << 32))
      return HF_E_INVALID;  // 32-bit unit offsets; the affine went into the split
    if constexpr (PG == 4) {
      if (wide) {
        if (P.vsplit) hipLaunchKernelGGL((gemm1x1_h<NTERMS, PG, true, true, 4>), grid, dim3(512), lds, st, P, wth, wtl);
        else hipLaunchKernelGGL((gemm1x1_h<NTERMS, PG, true, false, 4>), grid, dim3(512), lds, st, P, wth, wtl);
        return hf_launch_status();
      }
    }
    if (P.vsplit) hipLaunchKernelGGL((gemm1x1_h<NTERMS, PG, true, true>), grid, dim3(256), lds, st, P, wth, wtl);
    else hipLaunchKernelGGL((gemm1x1_h<NTERMS, PG, true>), grid, dim3(256), lds, st, P, wth, wtl);
  } else {
    if constexpr (PG == 4) {
      if (wide) {
        if (P.vsplit) hipLaunchKernelGGL((gemm1x1_h<NTERMS, PG, false, true, 4>), grid, dim3(512), lds, st, P, wth, wtl);
        else hipLaunchKernelGGL((gemm1x1_h<NTERMS, PG, false, false, 4>), grid, dim3(512), lds, st, P, wth, wtl);
        return hf_launch_status();
      }
    }
    if (P.vsplit) hipLaunchKernelGGL((gemm1x1_h<NTERMS, PG, false, true>), grid, dim3(256), lds, st, P, wth, wtl);
    else hipLaunchKernelGGL((gemm1x1_h<NTERMS, PG, false>), grid, dim3(256), lds, st, P, wth, wtl);
  }
  return hf_launch_status();
}

}  // namespace

// blocks of the launch (the tile form launch_gemm takes for this batch)
static long long gemm_blocks(int batch, int cout, int oplane, int groups) {
  const int pt = oplane * (long long)batch <= 128 || oplane <= 128 ? 128 : 256;
  int tw = pt, nb = 1;
  if (oplane < pt && (oplane & (oplane - 1)) == 0) {
    tw = oplane;
    nb = pt / oplane;
    if (nb > pow2_ceil(batch)) nb = pow2_ceil(batch);
  }
  return (long long)hf_cdiv(oplane, tw) * hf_cdiv(batch, nb) * (cout / 64) * (groups > 1 ? groups : 1);
}

// Split-K plan: small grids split K.  Batch-invariant mode: the K partition of the CANONICAL launch (kCanonBatch)...
static int gemm_splits(int batch, int cin, int cout, int oplane, int groups) {
  const long long blocks = gemm_blocks(hf_detail::plan_batch(batch), cout, oplane, groups);
  const int stages = cin / KS;
  int sk = 1;
  if (blocks < 256 && stages >= 4) {
    sk = (int)((512 + blocks - 1) / blocks);     // ~2 blocks per CU
    if (sk > stages / 2) sk = stages / 2;        // at least two stages per split (the pipeline's prologue)
    if (sk < 1) sk = 1;
  }
  return sk;
}
// ... and whether THIS launch runs it virtually (ConvParams::vsplit: its own output grid fills the chip; launch_gemm's rule)
static bool gemm_vsplit(int batch, int cout, int oplane, int groups, int sk) {
  return hf_detail::g_batch_invariant && sk > 1 && gemm_blocks(batch, cout, oplane, groups) >= gemm_fill_blocks();
}

extern "C" long long hf_conv1x1_f16_workspace_floats(int batch, int cin, int cout, int h, int w, int stride, int groups) {
  if (batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0 || (stride != 1 && stride != 2) || (cin % KS) || (cout % 64)) return 0;
  const int oh = (h - 1) / stride + 1, ow = (w - 1) / stride + 1;
  const int sk = gemm_splits(batch, cin, cout, oh * ow, groups);
  if (gemm_vsplit(batch, cout, oh * ow, groups, sk)) return 0;  // the blocks walk the slabs themselves: no slabs in memory
  return sk > 1 ? (long long)sk * (groups > 1 ? groups : 1) * batch * cout * oh * ow : 0;
}

extern "C" int hf_conv1x1_f16_f32(float *out, const float *x, const void *x_hi, const void *x_lo, const void *wt_hi,
                                  const void *wt_lo, int nterms, const float *in_scale, const float *in_shift, const float *out_scale, const float *bias,
                                  int act, const float *slope, float alpha, const float *residual, int batch, int cin,
                                  int cout, int h, int w, int stride, int groups, long long x_group_stride,
                                  float *workspace, long long workspace_floats, void *stream) {
  const int act_kind = act & ~HF_ACT_RESIDUAL_FIRST;
  if (!out || (!x && !x_hi) || !wt_hi || batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0 || (nterms != 1 && nterms != 3) ||
      (nterms == 3 && !wt_lo) || (stride != 1 && stride != 2) || (cin % KS) || (cout % 64) || act_kind < 0 || act_kind > ACT_QGELU ||
      (act_kind == ACT_PRELU && !slope) || groups < 1 || (groups > 1 && (in_scale || in_shift)) ||
      (x_hi && (in_scale || in_shift || stride != 1 || groups > 1 || (nterms == 3 && !x_lo))))
    return HF_E_INVALID;
  ConvParams P{};
  P.out = out; P.x = x; P.xh = x_hi; P.xl = x_lo; P.s = in_scale; P.t = in_shift; P.d = out_scale; P.bias = bias; P.slope = slope;
  P.residual = residual; P.residual_pre = (act & HF_ACT_RESIDUAL_FIRST) ? 1 : 0;
  P.s_bstride = 0; P.d_bstride = 0;
  P.batch = batch; P.cin = cin; P.cout = cout; P.h = h; P.w = w;
  P.stride = stride;
  const int oh = (h - 1) / stride + 1, ow = (w - 1) / stride + 1;
  // the epilogue (store_tile_rows) addresses an image's output as a 1 x (oh*ow) plane: out_w is the flat plane size,
  // the kernel recovers (oy, ox) with the true output width kept in out_wv
  P.out_h = 1; P.out_w = oh * ow; P.out_wv = ow;
  P.act = act_kind; P.alpha = alpha; P.scale = 1.0f;
  P.groups = groups; P.co_tiles = cout / 64; P.x_gstride = x_group_stride; P.wt_gstride = 0;
  if ((long long)batch * cin * h * w >= (1LL << 40)) return HF_E_INVALID;
  const int oplane = oh * ow;
  const int sk = gemm_splits(batch, cin, cout, oplane, groups);
  P.splits = sk;
  if (sk > 1) {
    const long long slab = (long long)(groups > 1 ? groups : 1) * batch * cout * oplane;
    if (!gemm_vsplit(batch, cout, oplane, groups, sk) && (!workspace || workspace_floats < sk * slab)) return HF_E_WORKSPACE;
    const int stages = cin / KS;
    P.chunks_per_split = (stages + sk - 1) / sk;
    P.splits = (stages + P.chunks_per_split - 1) / P.chunks_per_split;  // no empty split
    P.partial = workspace;
    P.zslab = slab;
  }
  const _Float16 *hi = static_cast<const _Float16 *>(wt_hi), *lo = static_cast<const _Float16 *>(wt_lo);
  const bool small = (long long)oplane * batch <= 128 || oplane <= 128;
  int rc;
  if (small) rc = (nterms == 3) ? launch_gemm<3, 2>(P, hi, lo, (hipStream_t)stream) : launch_gemm<1, 2>(P, hi, lo, (hipStream_t)stream);
  else rc = (nterms == 3) ? launch_gemm<3, 4>(P, hi, lo, (hipStream_t)stream) : launch_gemm<1, 4>(P, hi, lo, (hipStream_t)stream);
  if (rc != HF_OK) return rc;
  note_path(7, g_gemm_wide_last ? 6 : (small ? 1 : 2));
  if (P.splits > 1 && !P.vsplit && !P.counters) {
    P.out_h = 1; P.out_w = oplane;
    return launch_splitk_reduce(P, true, (hipStream_t)stream);
  }
  return HF_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Small-plane modulated 3x3 convolutions (the generator's 4^2 .. 16^2 tower: 512 -> 512 channels, 128 .. 2048 pixels per
// batch of 8): the nine taps as ONE 1x1 GEMM with M = 9*cout - y_tap[b, co, p] = sum_ci wt[tap][ci][co] * s[b,ci] x[b,ci,p],
// no shifts, no halo, no im2col - followed by a combine pass that adds the taps at their shifted positions (and the
// split-K partials, in a fixed order) and applies the layer's tail:
//   same resolution (ModulatedConv2d.forward, models/stylegan2/model.py:238-250, 273-277 + StyledConv's noise / bias /
//     leaky ReLU, :337-343):     out[y, x]  = act( d * sum_{ky,kx} y_(ky,kx)[y+ky-1, x+kx-1] + noise + bias )
//   transposed, stride 2 (:252-262, F.conv_transpose2d): tmp[2y+ky, 2x+kx] += d * y_(ky,kx)[y, x]  (gathered per output),
//     the (2h+1) x pitch intermediate hf_blur_noise_bias_act_* consume.
// The layers are weight-bound (9.4 MB per layer, 0.6 - 10 GFLOP): the GEMM spreads (tap, channel tile, K split) over the
// whole chip - the tiled conv kernels give a 16^2 layer at batch 8 only 64 blocks that each walk all of K.
__global__ __launch_bounds__(256) void small_combine(float *__restrict__ out, const float *__restrict__ y, int splits,
                                                     long long zslab, const float *__restrict__ d,
                                                     const float *__restrict__ noise, const float *__restrict__ noise_w,
                                                     long long noise_bstride, const float *__restrict__ bias, int batch,
                                                     int cout, int h, int w, int up, int out_h, int out_pitch, int out_wv,
                                                     float alpha, float scale) {
#pragma clang fp contract(on)  // see encoder_ops.hip: no cross-statement fusion in grid-stride loops
  const long long oplane = (long long)out_h * out_pitch, iplane = (long long)h * w;
  const long long total = (long long)batch * cout * oplane;
  const long long stride = (long long)gridDim.x * 256;
  const float nw = noise ? noise_w[0] : 0.0f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int ox = (int)(i % out_pitch);
    long long r = i / out_pitch;
    const int oy = (int)(r % out_h);
    r /= out_h;
    const int co = (int)(r % cout), b = (int)(r / cout);
    if (ox >= out_wv) continue;  // pitch padding of the transposed conv's intermediate
    float acc = 0.0f;
    const float *yb = y + ((long long)b * 9 * cout + co) * iplane;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        int sy, sx;
        bool ok;
        if (up) {
          const int ty = oy - ky, tx = ox - kx;
          ok = ty >= 0 && tx >= 0 && !(ty & 1) && !(tx & 1) && (ty >> 1) < h && (tx >> 1) < w;
          sy = ty >> 1;
          sx = tx >> 1;
        } else {
          sy = oy + ky - 1;
          sx = ox + kx - 1;
          ok = sy >= 0 && sy < h && sx >= 0 && sx < w;
        }
        if (ok) {
          const float *src = yb + (long long)(ky * 3 + kx) * cout * iplane + (long long)sy * w + sx;
          float v = 0.0f;
          for (int z = 0; z < splits; ++z) v += src[(long long)z * zslab];
          acc += v;
        }
      }
    float v = acc * (d ? d[(long long)b * cout + co] : 1.0f);
    if (!up) {
      if (noise) v = fmaf(nw, noise[(long long)b * noise_bstride + (long long)oy * out_pitch + ox], v);
      if (bias) v = hf_lrelu(v + bias[co], alpha, scale);
    }
    out[i] = v;
  }
}

typedef _Float16 hf_half8_g __attribute__((ext_vector_type(8)));
extern "C" int hf_split_activation_mod_f16(void *out_hi, void *out_lo, const float *x, const float *scale, long long images,
                                           int channels, int h, int w, void *stream);

static int small_gemm(ConvParams &P, const void *w9_hi, const void *w9_lo, int nterms, const float *x, const float *s,
                      int batch, int cin, int cout, int h, int w, float *workspace, long long workspace_floats,
                      hipStream_t st) {
  P = ConvParams{};
  // the modulated input, split once for all 9*cout/64 channel tiles: hi (+ lo) behind the tap slabs in the workspace
  const long long split_floats = (long long)batch * cin * h * w / 2;  // halves -> floats, per part
  {
    int sk0 = gemm_splits(batch, cin, 9 * cout, h * w, 1);
    if (gemm_vsplit(batch, 9 * cout, h * w, 1, sk0)) sk0 = 1;  // virtual split: one slab
    const long long need = (long long)sk0 * batch * 9 * cout * h * w + 2 * split_floats;
    if (!workspace || workspace_floats < need) return HF_E_WORKSPACE;
    float *xs_hi = workspace + (long long)sk0 * batch * 9 * cout * h * w, *xs_lo = xs_hi + split_floats;
    const int rc = hf_split_activation_mod_f16(xs_hi, nterms == 3 ? xs_lo : nullptr, x, s, batch, cin, h, w, st);
    if (rc != HF_OK) return rc;
    P.xh = xs_hi;
    P.xl = nterms == 3 ? xs_lo : nullptr;
  }
  P.x = x; P.s = nullptr; P.s_bstride = 0; P.d_bstride = 0;
  P.batch = batch; P.cin = cin; P.cout = 9 * cout; P.h = h; P.w = w; P.stride = 1;
  P.out_h = 1; P.out_w = h * w; P.out_wv = w;
  P.act = ACT_NONE; P.scale = 1.0f;
  P.groups = 1; P.co_tiles = 9 * cout / 64;
  const int oplane = h * w;
  int sk = gemm_splits(batch, cin, 9 * cout, oplane, 1);
  const int stages = cin / KS;
  P.chunks_per_split = (stages + sk - 1) / sk;
  sk = (stages + P.chunks_per_split - 1) / P.chunks_per_split;
  P.splits = sk;
  const long long slab = (long long)batch * 9 * cout * oplane;
  if (!workspace || workspace_floats < (long long)(gemm_vsplit(batch, 9 * cout, oplane, 1, sk) ? 1 : sk) * slab) return HF_E_WORKSPACE;
  // the GEMM always stores through the split-K path (raw sums into the workspace, one slab per split)
  P.partial = workspace;
  P.zslab = slab;
  P.out = workspace;
  if (sk == 1) P.splits = 1;  // one slab, written by the ordinary epilogue (no scale / bias / activation set)
  const _Float16 *hi = static_cast<const _Float16 *>(w9_hi), *lo = static_cast<const _Float16 *>(w9_lo);
  bool small = (long long)oplane * batch <= 128 || oplane <= 128;
  if (!small && !(g_h_tune & 2)) {
    // 256-pixel tiles run two blocks per CU (512 slots), 128-pixel tiles three (768; 148-152 registers, 48 KB): when the last
    // round of the 256-pixel form would be under half full, the launch takes the 128-pixel form - a tile FORM, same K order,
    // same bits.  The 16^2 layers at batch 8: 576 blocks on 512 slots -> 1152 on 768, 52 -> 44 us (profiles/r06am_*);
    // hf_debug_set_tuning bit 1 = never (A/B, tests)
    const long long n256 = (long long)hf_cdiv(oplane, 256) * batch * (9 * cout / 64) * (gemm_vsplit(batch, 9 * cout, oplane, 1, sk) ? 1 : sk);
    const long long slots = 2LL * gemm_fill_blocks(), tail = n256 % slots;
    small = n256 > slots && tail > 0 && 2 * tail <= slots;
  }
  if (small) return (nterms == 3) ? launch_gemm<3, 2>(P, hi, lo, st) : launch_gemm<1, 2>(P, hi, lo, st);
  return (nterms == 3) ? launch_gemm<3, 4>(P, hi, lo, st) : launch_gemm<1, 4>(P, hi, lo, st);
}

extern "C" long long hf_modconv3x3_small_workspace_floats(int batch, int cin, int cout, int h, int w) {
  if (batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0 || (cin % KS) || (cout % 64)) return 0;
  int sk = gemm_splits(batch, cin, 9 * cout, h * w, 1);
  if (gemm_vsplit(batch, 9 * cout, h * w, 1, sk)) sk = 1;
  return (long long)sk * batch * 9 * cout * h * w + (long long)batch * cin * h * w;  // tap slabs + the split input (hi, lo)
}

extern "C" int hf_modconv3x3_small_f16_f32(float *out, const float *x, const void *w9_hi, const void *w9_lo, int nterms,
                                           const float *s, const float *d, const float *noise, const float *noise_w,
                                           long long noise_bstride, const float *bias, int batch, int cin, int cout, int h,
                                           int w, float alpha, float scale, int upsample, int tmp_pitch, float *workspace,
                                           long long workspace_floats, void *stream) {
  if (!out || !x || !w9_hi || batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0 || (nterms != 1 && nterms != 3) ||
      (nterms == 3 && !w9_lo) || (cin % KS) || (cout % 64) || (noise && !noise_w) || h * w > 1024 ||
      (upsample && (tmp_pitch < 2 * w + 1 || noise || bias)))
    return HF_E_INVALID;
  ConvParams P;
  const int rc = small_gemm(P, w9_hi, w9_lo, nterms, x, s, batch, cin, cout, h, w, workspace, workspace_floats, (hipStream_t)stream);
  if (rc != HF_OK) return rc;
  const int out_h = upsample ? 2 * h + 1 : h, pitch = upsample ? tmp_pitch : w, wv = upsample ? 2 * w + 1 : w;
  const long long total = (long long)batch * cout * out_h * pitch;
  long long g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(small_combine, dim3((int)g), dim3(256), 0, (hipStream_t)stream, out, workspace, P.vsplit ? 1 : P.splits, P.zslab, d, noise,
                     noise_w, noise_bstride, bias, batch, cout, h, w, upsample ? 1 : 0, out_h, pitch, wv, alpha, scale);
  note_path(7, upsample ? 4 : 3);
  return hf_launch_status();
}


// ------------------------------------------------------------------------------------------------------------------
// Round 6: the small-plane UPSAMPLING StyledConv in two launches instead of four.  After the tap GEMM, ONE kernel per
// (image, 8-channel block) builds the demodulated (2h+1) x (2w+1) intermediate of small_combine's transposed form in LDS
// (with a zero border: the blur's pad (1,1)), blurs it and applies noise + bias + leaky ReLU, and writes either the fp32
// activation or - for a consumer on the fp16 matrix cores - s_next * out split into fp16 (hi, lo) K-blocked units.  The
// intermediate (17.8 MB at 16^2 -> 32^2, batch 8) never goes to memory: small_combine (69 us there) + blur4x4_split8 (27 us)
// become ~30 us.  Arithmetic = small_combine (up) followed by blur4x4_noise_bias_act / blur4x4_split8, operation for
// operation: bit-identical to the three-launch path (tests/test_sim_generator.py, tests/test_gpu_parity.py).
template <bool SPLIT>
__global__ __launch_bounds__(1024) void small_up_blur(float *__restrict__ out, hf_half8_g *__restrict__ out_hi,
                                                      hf_half8_g *__restrict__ out_lo, const float *__restrict__ y, int splits,
                                                      long long zslab, const float *__restrict__ d,
                                                      const float *__restrict__ kernel4x4, const float *__restrict__ noise,
                                                      const float *__restrict__ noise_w, long long noise_bstride,
                                                      const float *__restrict__ bias, const float *__restrict__ s_next, int cout,
                                                      int h, int w, float alpha, float scale) {
#pragma clang fp contract(on)  // as small_combine: no cross-statement fusion
  HF_DYN_LDS;
  const int th = 2 * h + 3, tw = 2 * w + 3, tplane = th * tw, iplane = h * w;
  float *T = reinterpret_cast<float *>(hf_dyn_lds);  // [8][2h+3][2w+3]: rows / columns -1 .. 2h+1 (2w+1) of the intermediate
  float *V = T + 8 * tplane;                         // [8][9][h*w]: the tap planes of the block's channels, K slabs added
  const int cblocks = cout >> 3;
  const int b = blockIdx.x / cblocks, cb = blockIdx.x - b * cblocks;
  constexpr int NT = 1024;  // sixteen waves: the block is alone on its CU (113 KB of LDS at 16 x 16) and its work is index
                            // arithmetic + latency - four waves per SIMD issue VALU at 2 instead of 3 cycles (tools/probes/valu_rate.hip)
  // Index walks without per-element divisions (a runtime division is ~40 VALU instructions: the first form of this kernel
  // spent most of its 65 us on them): (q, p) = (i / n, i % n) advanced by NT with carries.
  // ---- phase 0: V[ct][p] = sum_z y[z][b][tap*cout + co][p], ct = c*9 + tap, z ascending from 0 (small_combine's inner
  // loop); coalesced loads, four elements per thread in flight together (nine: the 4^2 / 8^2 planes 16 -> 23 us, the 16^2 plane
  // unchanged - r06ap)
  {
    constexpr int U = 4;
    const int qs = NT / iplane, ps = NT - qs * iplane;
    int ct = threadIdx.x / iplane, p = threadIdx.x - ct * iplane;
    while (ct < 72) {
      const float *src[U];
      float t[U];
      int vi[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int cc = min(ct, 71);                    // (clamped: surplus elements of the last pass re-read a valid one)
        const int c = cc / 9, tap = cc - c * 9;
        src[u] = y + ((long long)b * 9 * cout + (long long)tap * cout + cb * 8 + c) * iplane + p;
        vi[u] = ct < 72 ? ct * iplane + p : -1;
        t[u] = 0.0f;
        ct += qs;
        p += ps;
        if (p >= iplane) {
          p -= iplane;
          ++ct;
        }
      }
      for (int z = 0; z < splits; ++z) {
#pragma unroll
        for (int u = 0; u < U; ++u) t[u] += src[u][(long long)z * zslab];
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (vi[u] >= 0) V[vi[u]] = t[u];
    }
  }
  __syncthreads();
  // ---- phase 1: T[c][Y][X] = d * sum over the taps (ky, kx ascending) that reach (Y, X)  (small_combine, up = 1), zero outside
  {
    const int dy = NT / tw, dx = NT - dy * tw;
    int c = threadIdx.x / tplane, r = threadIdx.x - c * tplane;
    int ry = r / tw, rx = r - ry * tw;
    while (c < 8) {
      const int oy = ry - 1, ox = rx - 1;
      float v = 0.0f;
      if (oy >= 0 && oy <= 2 * h && ox >= 0 && ox <= 2 * w) {
        float acc = 0.0f;
        const float *vb = V + c * 9 * iplane;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int ty = oy - ky, tx = ox - kx;
            const bool ok = ty >= 0 && tx >= 0 && !(ty & 1) && !(tx & 1) && (ty >> 1) < h && (tx >> 1) < w;
            if (ok) acc += vb[(ky * 3 + kx) * iplane + (ty >> 1) * w + (tx >> 1)];
          }
        v = acc * (d ? d[(long long)b * cout + cb * 8 + c] : 1.0f);
      }
      T[c * tplane + ry * tw + rx] = v;
      rx += dx;
      ry += dy;
      if (rx >= tw) {
        rx -= tw;
        ++ry;
      }
      while (ry >= th) {
        ry -= th;
        ++c;
      }
    }
  }
  __syncthreads();
  // ---- phase 2: 4x4 blur (true convolution: flipped taps), noise, bias, leaky ReLU; tap order of blur4x4_noise_bias_act
  float kf[4][4];
#pragma unroll
  for (int ky = 0; ky < 4; ++ky)
#pragma unroll
    for (int kx = 0; kx < 4; ++kx) kf[ky][kx] = kernel4x4[(3 - ky) * 4 + (3 - kx)];
  const int out_h = 2 * h, out_w = 2 * w;
  const float nw = noise ? noise_w[0] : 0.0f;
  const float *nz = noise ? noise + (long long)b * noise_bstride : nullptr;
  bool ovf = false;
  for (int p = threadIdx.x; p < out_h * out_w; p += NT) {
    const int oy = p / out_w, ox = p - oy * out_w;
    const float nzr = nz ? nz[(long long)oy * out_w + ox] : 0.0f;
    float res[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float *win = T + c * tplane + oy * tw + ox;  // window rows oy-1 .. oy+2 = T rows oy .. oy+3 (border offset 1)
      float acc = 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = fmaf(win[r * tw + j], kf[r][j], acc);
      if (nz) acc = fmaf(nw, nzr, acc);
      if (bias) acc = hf_lrelu(acc + bias[cb * 8 + c], alpha, scale);
      res[c] = acc;
    }
    if (!SPLIT) {
#pragma unroll
      for (int c = 0; c < 8; ++c) out[(((long long)b * cout + cb * 8 + c) * out_h + oy) * out_w + ox] = res[c];
    } else {
      hf_half8_g h8, l8;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        float v = res[c] * (s_next ? s_next[(long long)b * cout + cb * 8 + c] : 1.0f);
        _Float16 hv, lv;
        hf_split_f16(v, hv, lv, ovf);
        h8[c] = hv;
        l8[c] = lv;
      }
      const long long unit = (((long long)b * cblocks + cb) * out_h + oy) * out_w + ox;
      out_hi[unit] = h8;
      if (out_lo) out_lo[unit] = l8;
    }
  }
  if (SPLIT) hf_note_overflow(ovf);
}

extern "C" int hf_modconv3x3_small_up_blur_f16_f32(float *out, void *out_hi, void *out_lo, const float *x, const void *w9_hi,
                                                   const void *w9_lo, int nterms, const float *s, const float *d,
                                                   const float *blur_kernel4x4, const float *noise, const float *noise_w,
                                                   long long noise_bstride, const float *bias, const float *s_next, int batch,
                                                   int cin, int cout, int h, int w, float alpha, float scale, float *workspace,
                                                   long long workspace_floats, void *stream) {
  if ((!out == !out_hi) || !x || !w9_hi || !blur_kernel4x4 || batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0 ||
      (nterms != 1 && nterms != 3) || (nterms == 3 && !w9_lo) || (cin % KS) || (cout % 64) || (noise && !noise_w) || h * w > 1024 ||
      (long long)batch * (cout >> 3) > 0x7fffffffLL)
    return HF_E_INVALID;  // exactly one output form
  const size_t lds = ((size_t)8 * (2 * h + 3) * (2 * w + 3) + (size_t)72 * h * w) * sizeof(float);  // T + the tap planes
  if (lds > 150 * 1024) return HF_E_INVALID;  // (16 x 16 inputs: 113 KB; the generator's small-plane layers stop there)
  ConvParams P;
  const int rc = small_gemm(P, w9_hi, w9_lo, nterms, x, s, batch, cin, cout, h, w, workspace, workspace_floats, (hipStream_t)stream);
  if (rc != HF_OK) return rc;
  const int splits = P.vsplit ? 1 : P.splits;
  const dim3 grid(batch * (cout >> 3));
  if (out_hi)
    hipLaunchKernelGGL(small_up_blur<true>, grid, dim3(1024), lds, (hipStream_t)stream, nullptr, static_cast<hf_half8_g *>(out_hi),
                       static_cast<hf_half8_g *>(out_lo), workspace, splits, P.zslab, d, blur_kernel4x4, noise, noise_w, noise_bstride,
                       bias, s_next, cout, h, w, alpha, scale);
  else
    hipLaunchKernelGGL(small_up_blur<false>, grid, dim3(1024), lds, (hipStream_t)stream, out, nullptr, nullptr, workspace, splits,
                       P.zslab, d, blur_kernel4x4, noise, noise_w, noise_bstride, bias, nullptr, cout, h, w, alpha, scale);
  note_path(7, 5);
  return hf_launch_status();
}

extern "C" unsigned long long hf_f16_overflow_count_gemm(int reset) { return hf_f16_overflow_read_tu(reset); }
