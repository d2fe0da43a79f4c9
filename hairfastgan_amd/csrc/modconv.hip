// modconv.hip - fp32 MFMA implicit-GEMM convolution kernels.
//
// Generator (reference models/stylegan2/model.py):
//   hf_modconv3x3_f32     same-resolution StyledConv: ModulatedConv2d (+demod) +
//                         NoiseInjection + FusedLeakyReLU  (:238-250, :273-277, :337-343)
//   hf_modconv3x3_up_f32  upsampling ModulatedConv2d, conv_transpose2d(stride 2) part
//                         (:252-262); blur + noise + activation: hf_blur_noise_bias_act_f32.
// Encoders (e4e: models/encoder4editing/models/encoders/helpers.py:93-120, psp_encoders.py:34-55;
// FeatureStyle: models/FeatureStyleEncoder/arcface/iresnet.py:28-57):
//   hf_conv2d_f32         Conv2d 3x3 / 1x1, stride 1 / 2, with the surrounding BatchNorm2d
//                         (inference statistics) folded in: a per-input-channel affine applied
//                         to the activations while they are staged (the pre-conv BN cannot be
//                         folded into the weights because of the zero padding), a per-output-
//                         channel scale + bias (post-conv BN), PReLU / LeakyReLU and the
//                         residual add in the epilogue.
//
// Formulation (MI355X-first, not the reference's per-sample grouped conv):
//   y[b,co,p] = d[b,co] * sum_{tap,ci} wt[tap,ci,co] * (s[b,ci] * x[b,ci,p+tap] + t[ci])
// i.e. the modulation s is applied to the ACTIVATIONS while they are staged
// into LDS and the demodulation d to the OUTPUTS in the epilogue, so one weight
// tensor is shared by the whole batch: an implicit GEMM with M = cout,
// N = batch*H*W pixels, K = taps*cin, run on v_mfma_f32_32x32x2_f32 (exact fp32,
// bitwise an fmaf chain; 157 TFLOP/s peak = the roofline of these layers).
//
// Transposed conv without wasted zero-insert flops: output (2Y+pr, 2X+pc) only
// receives taps with ky = pr, kx = pc (mod 2), so each of the 9 taps feeds one
// of 4 phase accumulators: same MFMA count as a 3x3 conv at INPUT resolution.
//
// Work decomposition: block = 4 or 8 waves (64 lanes each); block tile = CT
// output channels x PT pixels.  Pixels of a tile are (image, row, col) triples
// described at run time by a TileGeom, so one kernel serves 1024^2 planes (8x32
// tiles), 4x4 planes (8 whole images per tile) and the rim tiles of the
// (2H+1)x(2W+1) transposed-conv output.  K loop: chunks of KC=8 input channels
// staged in LDS as wt[taps][KC][CT] and the halo tile x[KC][images][rows][cols];
// every wave issues taps*KC/2 MFMAs per (co tile, pixel group), reading one fp32 A
// and B operand per lane with conflict-free ds_read_b32 (A: 32 consecutive co,
// B: 32 consecutive pixels; lanes 32-63 take the next input channel).
#include "conv_common.h"

using namespace hf_detail;

namespace {

struct NoSideWork {
  __device__ __forceinline__ void operator()(int) const {}
};

// One K chunk: TAPS x KC/2 steps of v_mfma_f32_32x32x2_f32 per (co tile, pixel group).
// a_base: &wl[lh][wave_co + li] (A operand: 32 consecutive co, lanes 32-63 the next ci);
// b_base: &xl[lh * xstride]      (B operand: the lane's pixel, shifted per tap).
// Software pipelined: the A/B fragments of step n+1 are read from LDS before the MFMAs
// of step n issue, so the ~100-cycle ds_read latency sits under 64*CT_TILES*PG cycles of
// matrix-pipe work.  `side(step)` is called once per step (step is a compile-time
// constant after unrolling) between the operand prefetch and the step's MFMAs: the
// pipelined kernel uses it to spread its staging instructions (weight DMA, halo loads,
// LDS writes) over the chunk, so that they issue under matrix-pipe time.
template <int CT_TILES, int PG, int CT, bool UP, int TAPS, class BLoad, class Side, int NPH = (UP ? 4 : 1)>
__device__ __forceinline__ void mfma_chunk_g(f32x16 (&acc)[NPH][CT_TILES][PG], const float *a_base, BLoad &&bload,
                                             Side &&side) {
  constexpr int NSTEP = TAPS * (KC / 2);
  float a[2][CT_TILES], bq[2][PG];
  auto fetch = [&](int step, float (&av)[CT_TILES], float (&bv)[PG]) {
    const int tap = step / (KC / 2), kk = step % (KC / 2);
#pragma unroll
    for (int ct = 0; ct < CT_TILES; ++ct) av[ct] = a_base[(tap * KC + 2 * kk) * CT + ct * 32];
#pragma unroll
    for (int g = 0; g < PG; ++g) bv[g] = bload(tap, kk, g);
  };
  fetch(0, a[0], bq[0]);
#pragma unroll
  for (int step = 0; step < NSTEP; ++step) {
    const int cur = step & 1;
    if (step + 1 < NSTEP) fetch(step + 1, a[cur ^ 1], bq[cur ^ 1]);
    side(step);
    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of this step's MFMAs
    const int tap = step / (KC / 2);
    const int ph = UP ? (((tap / 3) & 1) * 2 + ((tap % 3) & 1)) : 0;
#pragma unroll
    for (int ct = 0; ct < CT_TILES; ++ct)
#pragma unroll
      for (int g = 0; g < PG; ++g)
        acc[ph][ct][g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][ct], bq[cur][g], acc[ph][ct][g], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// B operand from a rectangular halo tile: b_base[2kk*xstride + pixoff[g] + tap offset].
// UP: output phase (pr,pc) = (ky&1, kx&1) reads x[Y - (ky==2), X - (kx==2)]; the halo tile
// starts at (-1,-1).
template <int CT_TILES, int PG, int CT, bool UP, int TAPS, class Side, int NPH = (UP ? 4 : 1)>
__device__ __forceinline__ void mfma_chunk(f32x16 (&acc)[NPH][CT_TILES][PG], const float *a_base,
                                           const float *b_base, int xstride, const int (&pixoff)[PG], int wp,
                                           Side &&side) {
  auto bload = [&](int tap, int kk, int g) {
    constexpr int KW = (TAPS == 49) ? 7 : 3;  // square kernels: 1x1, 3x3, 7x7 (BiSeNet's ResNet stem)
    const int ky = tap / KW, kx = tap % KW;
    const int toff = (TAPS == 1) ? 0 : (UP ? ((ky == 2 ? 0 : 1) * wp + (kx == 2 ? 0 : 1)) : (ky * wp + kx));
    return b_base[2 * kk * xstride + pixoff[g] + toff];
  };
  mfma_chunk_g<CT_TILES, PG, CT, UP, TAPS>(acc, a_base, bload, side);
}

// ------------------------------------------------------------------------------
// General kernel: any channel counts, multi-image tiles, stride 1/2, 3x3 / 1x1,
// split-K first pass.  Synchronous staging (load -> LDS -> barrier -> MFMA).
// CT_TILES x PG 32x32 MFMA tiles per wave; WAVES_CO x WAVES_PX = 4 waves.
// ------------------------------------------------------------------------------
template <int CT_TILES, int PG, int WAVES_CO, int WAVES_PX, bool UP, int TAPS>
__global__ __launch_bounds__(kThreads) void conv_mfma(const ConvParams P) {
  const GroupOfs go = group_offsets(P);
  const int co_tile = go.co_tile;
  static_assert(WAVES_CO * WAVES_PX == 4, "4 waves per block");
  static_assert(!(UP && TAPS != 9), "transposed conv is 3x3");
  constexpr int CT = 32 * CT_TILES * WAVES_CO;
  constexpr int NPH = UP ? 4 : 1;
  constexpr int KH = (TAPS == 49) ? 7 : ((TAPS == 9) ? 3 : 1);

  HF_DYN_LDS;
  float *wl = reinterpret_cast<float *>(hf_dyn_lds);  // [TAPS][KC][CT]
  float *xl = wl + TAPS * KC * CT;                    // [KC][xs_max]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int li = lane & 31;  // MFMA row (A) / column (B) index
  const int lh = lane >> 5;  // MFMA k index within the k=2 step
  const int wave_co = (wave / WAVES_PX) * (32 * CT_TILES);
  const int wave_pg = (wave % WAVES_PX) * PG;
  const int co0 = co_tile * CT;

  // ---- which tile family / tile is this block? (uniform) --------------------
  int gi = 0;
  if (P.n_geom > 1 && (int)blockIdx.x >= P.g[1].first_block) gi = 1;
  if (P.n_geom > 2 && (int)blockIdx.x >= P.g[2].first_block) gi = 2;
  const TileGeom G = P.g[gi];
  int t = blockIdx.x - G.first_block;
  const int tx = t % G.tiles_x;
  t /= G.tiles_x;
  const int ty = t % G.tiles_y;
  const int tb = t / G.tiles_y;
  const int tw = 1 << G.lg_tw, th = 1 << G.lg_th, nb = 1 << G.lg_nb;
  const int st = UP ? 1 : P.stride;
  // input rows / cols a tile needs: UP x[Y-1..Y]; otherwise (th-1)*stride + KH
  const int wp = UP ? tw + 1 : (tw - 1) * st + KH;
  const int hp = UP ? th + 1 : (th - 1) * st + KH;
  const int pad = UP ? 1 : KH / 2;
  const int xs = nb * hp * wp;         // LDS floats per staged channel (<= xs_max)
  const int ty0 = G.y0 + ty * th;      // first output pixel row / col of the tile
  const int tx0 = G.x0 + tx * tw;
  const int b0 = tb * nb;
  const long long plane = (long long)P.h * P.w;
  const float *xb = P.x + go.x + (long long)b0 * P.cin * plane;

  // ---- per-thread staging descriptors for the halo tile (chunk invariant) ----
  int e_ofs[kMaxElemPerCi];  // float offset from xb (channel 0), -1 = zero fill
  int e_img[kMaxElemPerCi];  // image index within the tile (for s)
#pragma unroll
  for (int e = 0; e < kMaxElemPerCi; ++e) {
    const int idx = tid + e * kThreads;
    e_ofs[e] = -1;
    e_img[e] = 0;
    if (idx < xs) {
      const int im = idx / (hp * wp);
      const int rem = idx - im * (hp * wp);
      const int hy = rem / wp;
      const int hx = rem - hy * wp;
      const int ys = ty0 * st + hy - pad, xc = tx0 * st + hx - pad;
      if (ys >= 0 && ys < P.h && xc >= 0 && xc < P.w && b0 + im < P.batch) {
        e_ofs[e] = (int)((long long)im * P.cin * plane + (long long)ys * P.w + xc);
        e_img[e] = im;
      }
    }
  }

  // ---- per-lane pixel bookkeeping ------------------------------------------------
  int pixoff[PG];  // LDS offset of the pixel inside one channel's halo tile (tap (0,0))
#pragma unroll
  for (int g = 0; g < PG; ++g) {
    const int p = (wave_pg + g) * 32 + li;
    const int px = p & (tw - 1);
    const int py = (p >> G.lg_tw) & (th - 1);
    const int im = p >> (G.lg_tw + G.lg_th);
    pixoff[g] = im * hp * wp + py * st * wp + px * st;
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

  const bool cout_vec4 = (P.cout & 3) == 0;
  const float *a_base = wl + lh * CT + wave_co + li;
  const float *b_base = xl + lh * P.xs_max;

  const int ci_begin = (P.splits > 1) ? (int)blockIdx.z * P.chunks_per_split * KC : 0;
  const int ci_end = (P.splits > 1) ? min(P.cin, ci_begin + P.chunks_per_split * KC) : P.cin;
  for (int ci0 = ci_begin; ci0 < ci_end; ci0 += KC) {
    __syncthreads();  // previous chunk fully consumed

    // ---- stage weights: wl[tap][kc][c] = wt[tap][ci0+kc][co0+c] ---------------
    if (cout_vec4) {
      constexpr int C4 = CT / 4;
      for (int i = tid; i < TAPS * KC * C4; i += kThreads) {
        const int c4 = i % C4;
        const int kc = (i / C4) % KC;
        const int tap = i / (C4 * KC);
        const int ci = ci0 + kc, co = co0 + c4 * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ci < P.cin && co < P.cout)
          v = *reinterpret_cast<const float4 *>(P.wt + go.wt + ((long long)tap * P.cin + ci) * P.cout + co);
        *reinterpret_cast<float4 *>(wl + (tap * KC + kc) * CT + c4 * 4) = v;
      }
    } else {
      for (int i = tid; i < TAPS * KC * CT; i += kThreads) {
        const int c = i % CT;
        const int kc = (i / CT) % KC;
        const int tap = i / (CT * KC);
        const int ci = ci0 + kc, co = co0 + c;
        float v = 0.f;
        if (ci < P.cin && co < P.cout) v = P.wt[go.wt + ((long long)tap * P.cin + ci) * P.cout + co];
        wl[(tap * KC + kc) * CT + c] = v;
      }
    }

    // ---- stage the halo tile: xl[kc][idx] = s[b,ci] * x[b,ci,ys,xc] + t[ci] (0 outside) ---
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      const int ci = ci0 + kc;
      const bool cv = ci < P.cin;
#pragma unroll
      for (int e = 0; e < kMaxElemPerCi; ++e) {
        const int idx = tid + e * kThreads;
        if (idx < xs) {
          float v = 0.f;
          if (cv && e_ofs[e] >= 0) {
            v = xb[(long long)ci * plane + e_ofs[e]];
            if (P.s) v *= P.s[(long long)(b0 + e_img[e]) * P.s_bstride + ci];
            if (P.t) v += P.t[ci];
          }
          xl[kc * P.xs_max + idx] = v;
        }
      }
    }
    __syncthreads();

    mfma_chunk<CT_TILES, PG, CT, UP, TAPS>(acc, a_base, b_base, P.xs_max, pixoff, wp, NoSideWork());
  }

  if constexpr (!UP) {
    // whole channel tiles of an encoder-type launch (per-channel scale / bias, no noise): the epilogue without
    // per-element switches (conv_common.h)
    if (!P.noise && P.d_bstride == 0 && co0 + CT <= P.cout)
      store_tile_rows<CT_TILES, PG>(P, G, go, acc, co0 + wave_co, wave_pg, li, lh, ty0, tx0, b0);
    else
      store_tile<CT_TILES, PG, UP>(P, G, go, acc, co0 + wave_co, wave_pg, li, lh, ty0, tx0, b0);
    // split-K without a second launch: the tile's last block adds the slabs and runs the epilogue (conv_common.h)
    if (P.splits > 1 && P.counters && splitk_arrive_last(P, reinterpret_cast<int *>(hf_dyn_lds)))
      store_tile<CT_TILES, PG, false, 1, true>(P, G, go, acc, co0 + wave_co, wave_pg, li, lh, ty0, tx0, b0);
    return;
  }
  store_tile<CT_TILES, PG, UP>(P, G, go, acc, co0 + wave_co, wave_pg, li, lh, ty0, tx0, b0);
}

// ------------------------------------------------------------------------------
// Pipelined variant for the large 3x3 stride-1 layers (the >95 % of the FLOPs).
//
// Preconditions (checked on the host): cin % KC == 0, cout % CT == 0, every tile
// family holds a single image per tile (lg_nb == 0), xs <= XEP * NT.
// Double-buffered LDS: while the waves run the MFMAs of chunk c out of buffer
// c&1, the weights of chunk c+1 stream straight into the other buffer with
// global_load_lds_dwordx4 (1 KiB per wave instruction, no VGPRs) and the halo
// tile of chunk c+1 is prefetched into registers; it is multiplied by the
// (block-uniform) input scale and written to LDS late in the chunk, then one
// barrier per chunk.  The staging instructions are spread one per MFMA step.
// ABLATE (timing experiments only, results are wrong when != 0): 1 = no staging traffic
// inside the loop (barriers kept), 2 = no staging and no barriers.
// ------------------------------------------------------------------------------
template <int CT_TILES, int PG, int WAVES_CO, int WAVES_PX, bool UP, int ABLATE = 0, int STRIDE = 1>
__global__ __launch_bounds__(64 * WAVES_CO * WAVES_PX) void conv_mfma_pipe(const ConvParams P) {
  static_assert(!(UP && STRIDE != 1), "the transposed conv has no strided form");
  const GroupOfs go = group_offsets(P);
  const int co_tile = go.co_tile;
  constexpr int NW = WAVES_CO * WAVES_PX;
  constexpr int NT = 64 * NW;
  constexpr int CT = 32 * CT_TILES * WAVES_CO;
  constexpr int PT = 32 * PG * WAVES_PX;
  constexpr int NPH = UP ? 4 : 1;
  constexpr int HALO = UP ? 1 : 2;
  // halo elements per thread per ci: a tile of PT/32 rows x 32 output pixels needs
  // ((rows-1)*STRIDE + 1 + HALO) x (31*STRIDE + 1 + HALO) inputs
  constexpr int XEP = (((PT / 32 - 1) * STRIDE + 1 + HALO) * (31 * STRIDE + 1 + HALO) + NT - 1) / NT;
  constexpr int WCHUNK = 9 * KC * CT;                                   // floats of one weight stage
  constexpr int NDMA = WCHUNK / 256;                                    // 1 KiB wave-instructions per stage
  static_assert(WCHUNK % 256 == 0, "weight stage must be a whole number of 1 KiB DMA pieces");

  HF_DYN_LDS;
  float *wl0 = reinterpret_cast<float *>(hf_dyn_lds);  // [2][9][KC][CT]
  float *xl0 = wl0 + 2 * WCHUNK;                        // [2][KC][xs_max]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int li = lane & 31;
  const int lh = lane >> 5;
  const int wave_co = (wave / WAVES_PX) * (32 * CT_TILES);
  const int wave_pg = (wave % WAVES_PX) * PG;
  const int co0 = co_tile * CT;

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
  const int wp = (tw - 1) * STRIDE + 1 + HALO, hp = (th - 1) * STRIDE + 1 + HALO;
  const int xs = hp * wp;
  const int ty0 = G.y0 + ty * th;  // first OUTPUT pixel of the tile
  const int tx0 = G.x0 + tx * tw;
  const long long plane = (long long)P.h * P.w;
  const float *xb = P.x + go.x + (long long)b0 * P.cin * plane;
  const float *sb = P.s ? P.s + (long long)b0 * P.s_bstride : nullptr;

  int e_ofs[XEP];  // offset of the thread's halo elements inside a channel plane, -1 = zero
#pragma unroll
  for (int e = 0; e < XEP; ++e) {
    const int idx = tid + e * NT;
    e_ofs[e] = -1;
    if (idx < xs) {
      const int hy = idx / wp, hx = idx - hy * wp;
      const int ys = ty0 * STRIDE + hy - 1, xc = tx0 * STRIDE + hx - 1;
      if (ys >= 0 && ys < P.h && xc >= 0 && xc < P.w) e_ofs[e] = ys * P.w + xc;
    }
  }

  int pixoff[PG];
#pragma unroll
  for (int g = 0; g < PG; ++g) {
    const int p = (wave_pg + g) * 32 + li;
    pixoff[g] = ((p >> G.lg_tw) & (th - 1)) * STRIDE * wp + (p & (tw - 1)) * STRIDE;
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

  // Staging work items of one chunk (per thread): ND weight-DMA pieces, NL halo loads
  // and NL LDS writes.  Piece j of the DMA covers floats [256 j, 256 j + 256) of the stage
  // image [tap][kc][c]; a lane moves 4 floats of row (tap,kc) = f / CT at column f % CT.
  constexpr int ND = (NDMA + NW - 1) / NW;
  constexpr int NL = KC * XEP;
  auto dma_piece = [&](int i, int ci0, float *wl) {  // i-th piece of this wave
    const int j = wave + i * NW;
    if (j < NDMA) {
      const int f = j * 256 + lane * 4;
      const int row = f / CT, col = f % CT;
      const int tap = row / KC, kc = row % KC;
      hf_glds16(P.wt + go.wt + ((long long)tap * P.cin + ci0 + kc) * P.cout + co0 + col, wl + j * 256);
    }
  };
  float xr[KC][XEP];
  float sr[KC], tr[KC];  // input affine of the prefetched channels (fetched with the halo)
  auto load_piece = [&](int i, int ci0) {
    const int kc = i / XEP, e = i % XEP;
    xr[kc][e] = (e_ofs[e] >= 0) ? xb[(long long)(ci0 + kc) * plane + e_ofs[e]] : 0.0f;
    if (e == 0) {
      sr[kc] = sb ? sb[ci0 + kc] : 1.0f;
      tr[kc] = P.t ? P.t[ci0 + kc] : 0.0f;
    }
  };
  auto write_piece = [&](int i, float *xl) {
    const int kc = i / XEP, e = i % XEP;
    const int idx = tid + e * NT;
    if (idx < xs) xl[kc * P.xs_max + idx] = (e_ofs[e] >= 0) ? fmaf(xr[kc][e], sr[kc], tr[kc]) : 0.0f;
  };

  // split-K: this block handles chunks [c_begin, c_end) and stores raw partial sums
  const int nchunks = P.cin / KC;
  const int c_begin = (P.splits > 1) ? (int)blockIdx.z * P.chunks_per_split : 0;
  const int c_end = (P.splits > 1) ? min(nchunks, c_begin + P.chunks_per_split) : nchunks;

  // prologue: first chunk into buffer 0
#pragma unroll
  for (int i = 0; i < ND; ++i) dma_piece(i, c_begin * KC, wl0);
#pragma unroll
  for (int i = 0; i < NL; ++i) load_piece(i, c_begin * KC);
#pragma unroll
  for (int i = 0; i < NL; ++i) write_piece(i, xl0);
  __syncthreads();

  // Slots of the 36-step chunk: DMA pieces first, halo loads next, LDS writes of the
  // prefetched halo last (>= 12 steps = ~6 k matrix-pipe cycles after the loads issued).
  constexpr int NSTEP = 9 * (KC / 2);
  constexpr int DPS = (ND + 7) / 8;                 // DMA pieces per step
  constexpr int D_STEPS = (ND + DPS - 1) / DPS;
  constexpr int LPS = (NL + 11) / 12;               // loads (and writes) per step
  constexpr int L_STEPS = (NL + LPS - 1) / LPS;
  constexpr int W_FIRST = NSTEP - L_STEPS;
  static_assert(D_STEPS + L_STEPS <= W_FIRST, "staging schedule does not fit the chunk");

  for (int c = c_begin; c < c_end; ++c) {
    const int cur = (c - c_begin) & 1;
    const bool more = (c + 1 < c_end) && ABLATE == 0;
    const int ci_next = (c + 1) * KC;
    float *wl_next = wl0 + (cur ^ 1) * WCHUNK;
    float *xl_next = xl0 + (cur ^ 1) * KC * P.xs_max;
    auto side = [&](int step) {
      if (!more) return;
      if (step < D_STEPS) {
#pragma unroll
        for (int q = 0; q < DPS; ++q)
          if (step * DPS + q < ND) dma_piece(step * DPS + q, ci_next, wl_next);
      } else if (step < D_STEPS + L_STEPS) {
#pragma unroll
        for (int q = 0; q < LPS; ++q)
          if ((step - D_STEPS) * LPS + q < NL) load_piece((step - D_STEPS) * LPS + q, ci_next);
      } else if (step >= W_FIRST) {
#pragma unroll
        for (int q = 0; q < LPS; ++q)
          if ((step - W_FIRST) * LPS + q < NL) write_piece((step - W_FIRST) * LPS + q, xl_next);
      }
    };
    const float *a_base = wl0 + cur * WCHUNK + lh * CT + wave_co + li;
    const float *b_base = xl0 + (cur * KC + lh) * P.xs_max;
    mfma_chunk<CT_TILES, PG, CT, UP, 9>(acc, a_base, b_base, P.xs_max, pixoff, wp, side);
    if (ABLATE < 2) __syncthreads();  // also drains the weight DMA (vmcnt) before anyone reads the other buffer
  }

  if constexpr (!UP) {
    // whole channel tiles of an encoder-type launch (per-channel scale / bias, no noise): the epilogue without
    // per-element switches (conv_common.h)
    if (!P.noise && P.d_bstride == 0 && co0 + CT <= P.cout)
      store_tile_rows<CT_TILES, PG>(P, G, go, acc, co0 + wave_co, wave_pg, li, lh, ty0, tx0, b0);
    else
      store_tile<CT_TILES, PG, UP>(P, G, go, acc, co0 + wave_co, wave_pg, li, lh, ty0, tx0, b0);
    // split-K without a second launch: the tile's last block adds the slabs and runs the epilogue (conv_common.h)
    if (P.splits > 1 && P.counters && splitk_arrive_last(P, reinterpret_cast<int *>(hf_dyn_lds)))
      store_tile<CT_TILES, PG, false, 1, true>(P, G, go, acc, co0 + wave_co, wave_pg, li, lh, ty0, tx0, b0);
    return;
  }
  store_tile<CT_TILES, PG, UP>(P, G, go, acc, co0 + wave_co, wave_pg, li, lh, ty0, tx0, b0);
}

// ------------------------------------------------------------------------------
// DMA-staged variant of the pipelined kernel for same-resolution 3x3 layers whose
// planes are tiled by full 32-pixel rows (tw == 32, W % 4 == 0, no pre-conv shift).
//
// Per chunk the weights [9][KC][CT] and the interior of the halo tile [KC][hp][32] (rows of
// 128 B = 8 lanes x 16 B) arrive by global_load_lds into the other LDS buffer with no VGPR
// involved; out-of-image rows are never written and stay zero from a one-time clear.  Only
// the two edge columns (KC*hp*2 values, at most one per thread) go through a register.
// They are stored in a second array with the SAME row / channel strides as the interior,
// so a B fragment read is `base[kx][group] + immediate`: the per-lane base selects the
// interior (column px+kx-1) or, for lanes px=0 / px=31, the edge array - no address
// arithmetic in the MFMA loop.  The modulation s[b,ci] is applied to the B fragment after
// the LDS read (one v_mul per fragment; the same fp32 product as scaling before staging).
// ------------------------------------------------------------------------------
template <int CT_TILES, int PG, int WAVES_CO, int WAVES_PX, bool MOD>
__global__ __launch_bounds__(64 * WAVES_CO * WAVES_PX) void conv_mfma_dma(const ConvParams P) {
  constexpr int NW = WAVES_CO * WAVES_PX;
  constexpr int NT = 64 * NW;
  constexpr int CT = 32 * CT_TILES * WAVES_CO;
  constexpr int TH = PG * WAVES_PX;   // tile = TH rows x 32 columns
  constexpr int HP = TH + 2;          // staged rows
  constexpr int WCHUNK = 9 * KC * CT;
  constexpr int XI = KC * HP * 32;    // interior floats per stage; the edge array has the same shape
  constexpr int BUF = WCHUNK + 2 * XI;
  constexpr int N_W = WCHUNK / 256, N_XI = XI / 256;
  constexpr int NPIECE = N_W + N_XI;
  constexpr int ND = (NPIECE + NW - 1) / NW;  // DMA instructions per wave per chunk
  constexpr int NEDGE = KC * HP * 2;
  static_assert(WCHUNK % 256 == 0 && XI % 256 == 0, "stage images must be whole 1 KiB DMA pieces");
  static_assert(NEDGE <= NT, "one edge value per thread");

  HF_DYN_LDS;
  float *lds = reinterpret_cast<float *>(hf_dyn_lds);  // [2][BUF] then s[cin]
  float *sl = lds + 2 * BUF;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int li = lane & 31;
  const int lh = lane >> 5;
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

  // one-time clear of the halo regions of both buffers (+ modulation vector into LDS)
  for (int i = tid; i < 4 * XI; i += NT) lds[(i / (2 * XI)) * BUF + WCHUNK + i % (2 * XI)] = 0.0f;
  if (MOD)
    for (int i = tid; i < P.cin; i += NT) sl[i] = P.s[(long long)b0 * P.s_bstride + i];

  // per-lane DMA source offsets within channel ci0 (chunk invariant; -1: lane idle)
  int d_src[ND];
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    const int pc = wave + i * NW;
    d_src[i] = -1;
    if (pc >= N_W && pc < NPIECE) {  // interior: 8 rows of 8 x 16 B per piece
      const int r = (pc - N_W) * 8 + (lane >> 3), q = lane & 7;
      const int kc = r / HP, row = r % HP;
      const int ys = ty0 + row - 1, xc = tx0 + 4 * q;
      if (ys >= 0 && ys < P.h && xc < P.w) d_src[i] = (int)(kc * plane + (long long)ys * P.w + xc);
    }
  }
  auto dma_piece = [&](int i, int ci0, float *buf) {
    const int pc = wave + i * NW;
    if (pc < N_W) {
      const int f = pc * 256 + lane * 4;
      const int row = f / CT, col = f % CT;
      const int tap = row / KC, kc = row % KC;
      hf_glds16(P.wt + ((long long)tap * P.cin + ci0 + kc) * P.cout + co0 + col, buf + pc * 256);
    } else if (pc < NPIECE) {
      hf_glds16_if(d_src[i] >= 0, xb + (long long)ci0 * plane + d_src[i], buf + WCHUNK + (pc - N_W) * 256);
    }
  };
  // edge columns: thread e < NEDGE owns (kc, row, side)
  int e_src = -1, e_dst = 0;
  if (tid < NEDGE) {
    const int kc = tid / (HP * 2), row = (tid / 2) % HP, side = tid & 1;
    const int ys = ty0 + row - 1, xc = side ? tx0 + 32 : tx0 - 1;
    if (ys >= 0 && ys < P.h && xc >= 0 && xc < P.w) e_src = (int)(kc * plane + (long long)ys * P.w + xc);
    e_dst = WCHUNK + XI + (kc * HP + row) * 32 + side;
  }
  float e_val = 0.0f;

  // B-fragment base per (kx, group): interior column li+kx-1, or the edge array for li=0 / li=31
  int bbase[3][PG];
#pragma unroll
  for (int g = 0; g < PG; ++g) {
    const int rowbase = (wave_pg + g) * 32 + lh * (HP * 32);  // one 32-pixel row per pixel group
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int c = li + kx - 1;
      bbase[kx][g] = WCHUNK + rowbase + ((c >= 0 && c < 32) ? c : XI + (c < 0 ? 0 : 1));
    }
  }

  f32x16 acc[1][CT_TILES][PG];
#pragma unroll
  for (int ct = 0; ct < CT_TILES; ++ct)
#pragma unroll
    for (int g = 0; g < PG; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][ct][g][r] = 0.0f;

  __syncthreads();  // clear + s visible before the first DMA lands next to it
#pragma unroll
  for (int i = 0; i < ND; ++i) dma_piece(i, 0, lds);
  if (e_src >= 0) lds[e_dst] = xb[e_src];
  __syncthreads();  // drains vmcnt: stage 0 resident

  constexpr int NSTEP = 9 * (KC / 2);
  constexpr int DPS = (ND + NSTEP / 2 - 1) / (NSTEP / 2);  // DMA instructions per step (first half of the chunk)
  constexpr int D_STEPS = (ND + DPS - 1) / DPS;
  const int nchunks = P.cin / KC;
  for (int c = 0; c < nchunks; ++c) {
    const int cur = c & 1;
    const bool more = c + 1 < nchunks;
    const int ci0 = c * KC, ci_next = ci0 + KC;
    float *buf = lds + cur * BUF, *buf_next = lds + (cur ^ 1) * BUF;
    float sreg[KC / 2];
#pragma unroll
    for (int kk = 0; kk < KC / 2; ++kk) sreg[kk] = MOD ? sl[ci0 + 2 * kk + lh] : 1.0f;
    auto side = [&](int step) {
      if (!more) return;
      if (step < D_STEPS) {
#pragma unroll
        for (int q = 0; q < DPS; ++q)
          if (step * DPS + q < ND) dma_piece(step * DPS + q, ci_next, buf_next);
      } else if (step == D_STEPS) {
        if (e_src >= 0) e_val = xb[(long long)ci_next * plane + e_src];
      } else if (step == NSTEP - 1) {
        if (e_src >= 0) buf_next[e_dst] = e_val;
      }
    };
    auto bload = [&](int tap, int kk, int g) {
      const int ky = tap / 3, kx = tap % 3;
      const float v = buf[bbase[kx][g] + ky * 32 + 2 * kk * (HP * 32)];
      return MOD ? v * sreg[kk] : v;
    };
    const float *a_base = buf + lh * CT + wave_co + li;
    mfma_chunk_g<CT_TILES, PG, CT, false, 9>(acc, a_base, bload, side);
    __syncthreads();  // drains the DMA of the next stage (vmcnt) and frees the current one
  }

  store_tile<CT_TILES, PG, false>(P, G, GroupOfs{0, 0, 0, 0, 0}, acc, co0 + wave_co, wave_pg, li, lh, ty0, tx0, b0);
}

// Split-K second pass: out = epilogue(d * sum_z partial[z]) - deterministic (fixed z order).
__global__ __launch_bounds__(256) void splitk_reduce(const ConvParams P, long long slab, int with_epilogue) {
#pragma clang fp contract(on)  // no cross-statement fusion: remainder iterations of the grid-stride loop must round like the unrolled ones (see encoder_ops.hip)
  const long long oplane = (long long)P.out_h * P.out_w;
  const long long ovol = (long long)P.batch * P.cout * oplane;  // one group
  const float nw = (with_epilogue && P.noise) ? P.noise_w[0] : 0.0f;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < slab; i += stride) {
    const long long g = i / ovol;
    const long long pl = (i - g * ovol) / oplane;
    const int co = (int)(pl % P.cout);
    const long long b = pl / P.cout;
    // per-channel vectors of group g follow those of g-1
    P.out[i] = splitk_finish(P, i, b, g * P.cout + co, i % oplane, nw, with_epilogue != 0);
  }
}

// ------------------------------------------------------------------------------
// host side: tile-geometry selection
// ------------------------------------------------------------------------------

template <int CT_TILES, int PG, int WAVES_CO, int WAVES_PX, bool UP, int TAPS>
int launch_conv(ConvParams &P, hipStream_t st) {
  constexpr int CT = 32 * CT_TILES * WAVES_CO;
  constexpr int PT = 32 * PG * WAVES_PX;
  const int ext = UP ? 1 : (TAPS == 49 ? 6 : (TAPS == 9 ? 2 : 0));
  const int stride = UP ? 1 : P.stride;
  int nblocks = 0;
  if (UP) {
    // (2h+1)x(2w+1) output = phases of the (h+1)x(w+1) (Y,X) domain: interior h x w
    // tiles + the Y=h row (incl. corner) + the X=w column.
    P.n_geom = 3;
    P.g[0] = make_geom(0, 0, P.h, P.w, P.batch, PT, 0);
    nblocks = geom_blocks(P.g[0]);
    P.g[1] = make_geom(P.h, 0, 1, P.w + 1, P.batch, PT, nblocks);
    nblocks += geom_blocks(P.g[1]);
    P.g[2] = make_geom(0, P.w, P.h, 1, P.batch, PT, nblocks);
    nblocks += geom_blocks(P.g[2]);
  } else {
    P.n_geom = 1;
    P.g[0] = make_geom(0, 0, P.out_h, P.out_w, P.batch, PT, 0);
    nblocks = geom_blocks(P.g[0]);
  }
  int xs_max = 0;
  for (int i = 0; i < P.n_geom; ++i) xs_max = max(xs_max, geom_xs(P.g[i], stride, ext));
  if (xs_max > kMaxElemPerCi * kThreads) return HF_E_INVALID;
  for (int i = 0; i < P.n_geom; ++i)  // staging offsets are 32-bit, relative to the tile's first image
    if (((long long)P.cin << P.g[i].lg_nb) * P.h * P.w >= (1LL << 31)) return HF_E_INVALID;
  P.xs_max = (xs_max + 3) & ~3;
  const size_t lds = (size_t)(TAPS * KC * CT + KC * P.xs_max) * sizeof(float);
  if (P.splits < 1) P.splits = 1;
  P.co_tiles = hf_cdiv(P.cout, CT);
  P.zslab = (long long)max(1, P.groups) * P.batch * P.cout * P.out_h * P.out_w;
  dim3 grid(nblocks, P.co_tiles * max(1, P.groups), P.splits);
  if (grid.y > 65535) return HF_E_INVALID;
  P.counters = (!UP && P.splits > 1) ? splitk_counters_for((long long)grid.x * grid.y) : nullptr;
  hipLaunchKernelGGL((conv_mfma<CT_TILES, PG, WAVES_CO, WAVES_PX, UP, TAPS>), grid, dim3(kThreads), lds, st, P);
  return hf_launch_status();
}

// Pipelined launch (interior + rim families in one launch); returns HF_E_INVALID if the
// shape does not meet the kernel's preconditions (caller then uses the general kernel).
template <int CT_TILES, int PG, int WAVES_CO, int WAVES_PX, bool UP, int ABLATE = 0, int STRIDE = 1>
int launch_conv_pipe(ConvParams &P, hipStream_t st) {
  constexpr int NT = 64 * WAVES_CO * WAVES_PX;
  constexpr int CT = 32 * CT_TILES * WAVES_CO;
  constexpr int PT = 32 * PG * WAVES_PX;
  constexpr int HALO = UP ? 1 : 2;
  constexpr int XEP = (((PT / 32 - 1) * STRIDE + 1 + HALO) * (31 * STRIDE + 1 + HALO) + NT - 1) / NT;
  if (P.cin % KC || P.cout % CT || (P.cout & 3)) return HF_E_INVALID;
  if (!UP && P.stride != STRIDE) return HF_E_INVALID;
  if ((((size_t)P.wt) & 15) != 0 || (P.wt_gstride & 3)) return HF_E_INVALID;
  if (P.splits < 1) P.splits = 1;
  P.n_geom = 1;
  P.g[0] = UP ? make_geom(0, 0, P.h, P.w, P.batch, PT, 0) : make_geom(0, 0, P.out_h, P.out_w, P.batch, PT, 0);
  int nblocks = geom_blocks(P.g[0]);
  if (UP) {  // + the Y = h row (incl. corner) and the X = w column of the (h+1)x(w+1) phase domain
    P.n_geom = 3;
    P.g[1] = make_geom(P.h, 0, 1, P.w + 1, P.batch, PT, nblocks, true);
    nblocks += geom_blocks(P.g[1]);
    P.g[2] = make_geom(0, P.w, P.h, 1, P.batch, PT, nblocks, true);
    nblocks += geom_blocks(P.g[2]);
  }
  int xs = 0;
  for (int i = 0; i < P.n_geom; ++i) {
    if (P.g[i].lg_nb != 0) return HF_E_INVALID;  // the pipelined kernel wants one image per tile
    xs = max(xs, geom_xs(P.g[i], STRIDE, HALO));
  }
  if (xs > XEP * NT) return HF_E_INVALID;
  if ((long long)P.cin * P.h * P.w >= (1LL << 31)) return HF_E_INVALID;
  P.xs_max = (xs + 3) & ~3;
  const size_t lds = (size_t)2 * (9 * KC * CT + KC * P.xs_max) * sizeof(float);
  if (lds > 160 * 1024) return HF_E_INVALID;
  P.co_tiles = P.cout / CT;
  P.zslab = (long long)max(1, P.groups) * P.batch * P.cout * P.out_h * P.out_w;
  dim3 grid(nblocks, P.co_tiles * max(1, P.groups), P.splits);
  if (grid.y > 65535) return HF_E_INVALID;
  P.counters = (!UP && P.splits > 1) ? splitk_counters_for((long long)grid.x * grid.y) : nullptr;
  hipLaunchKernelGGL((conv_mfma_pipe<CT_TILES, PG, WAVES_CO, WAVES_PX, UP, ABLATE, STRIDE>), grid, dim3(NT), lds, st,
                     P);
  return hf_launch_status();
}

// All-DMA launch (same-resolution 3x3, rows of 32 pixels); HF_E_INVALID if the shape does not qualify.
template <int CT_TILES, int PG, int WAVES_CO, int WAVES_PX>
int launch_conv_dma(ConvParams &P, hipStream_t st) {
  constexpr int NT = 64 * WAVES_CO * WAVES_PX;
  constexpr int CT = 32 * CT_TILES * WAVES_CO;
  constexpr int TH = PG * WAVES_PX, HP = TH + 2;
  if (P.cin % KC || P.cout % CT || (P.cout & 3) || P.stride != 1 || P.t || P.groups > 1) return HF_E_INVALID;
  if ((P.w & 3) || P.w < 32 || P.h < TH) return HF_E_INVALID;
  if ((((size_t)P.wt) | ((size_t)P.x)) & 15) return HF_E_INVALID;
  if ((long long)P.cin * P.h * P.w >= (1LL << 31)) return HF_E_INVALID;
  P.splits = 1;
  P.n_geom = 1;
  TileGeom g{};
  g.y0 = 0; g.x0 = 0; g.dh = P.h; g.dw = P.w;
  g.lg_tw = 5; g.lg_th = ilog2(TH); g.lg_nb = 0;
  g.tiles_x = hf_cdiv(P.w, 32); g.tiles_y = hf_cdiv(P.h, TH); g.tiles_b = P.batch; g.first_block = 0;
  P.g[0] = g;
  const int XI = KC * HP * 32;
  if (KC * HP * 2 > NT) return HF_E_INVALID;
  const size_t lds = ((size_t)2 * (9 * KC * CT + 2 * XI) + (P.s ? P.cin : 0)) * sizeof(float);
  if (lds > 160 * 1024) return HF_E_INVALID;
  dim3 grid(geom_blocks(g), P.cout / CT);
  if (grid.y > 65535) return HF_E_INVALID;
  if (P.s)
    hipLaunchKernelGGL((conv_mfma_dma<CT_TILES, PG, WAVES_CO, WAVES_PX, true>), grid, dim3(NT), lds, st, P);
  else
    hipLaunchKernelGGL((conv_mfma_dma<CT_TILES, PG, WAVES_CO, WAVES_PX, false>), grid, dim3(NT), lds, st, P);
  return hf_launch_status();
}

// Split-K plan (pure function of the shape, so that callers can size the workspace).
// Regime 1: fewer than one 64x64 tile per CU -> split until ~2 blocks per CU.
// Regime 2: 1-3 tiles per CU and a long K loop: such layers are bound by the per-chunk
// round trip (global load -> LDS -> barrier), not by MFMA time; splitting K puts 3 blocks
// on every CU so that their round trips overlap.
inline int splitk_plan(int batch, int cin, int cout, int out_h, int out_w, bool allow_mid = true) {
  const int PT = 64, CT = 64;  // the small-plane configuration <1,1,2,2>
  const long long per_img = ((long long)out_h * out_w + PT - 1) / PT;
  const long long base = plan_batch(batch) * per_img * ((cout + CT - 1) / CT);
  const int nchunks = (cin + KC - 1) / KC;
  int s = 1;
  if (base < 256) {
    s = (int)((512 + base - 1) / base);
    if (s > nchunks) s = nchunks;
    if (s > 64) s = 64;
  } else if (allow_mid && base < 768 && nchunks >= 16) {
    s = (int)((768 + base - 1) / base);
    if (s > nchunks / 4) s = nchunks / 4;
    if (s > 8) s = 8;
  }
  return s < 2 ? 1 : s;
}

}  // namespace
// Arrival counters of the in-kernel split-K reduction: a caller-owned, zero-initialised device buffer registered per
// thread (hf_set_splitk_counters); every launch leaves it zero again.  NULL / too small: the two-launch form.
namespace hf_detail {
thread_local unsigned int *g_splitk_counters = nullptr;
thread_local int g_splitk_counter_ints = 0;
unsigned int *splitk_counters_for(long long tiles) {
  return (g_splitk_counters && tiles > 0 && tiles <= g_splitk_counter_ints) ? g_splitk_counters : nullptr;
}
}  // namespace hf_detail
extern "C" int hf_set_splitk_counters(void *zeroed_ints, int n_ints) {
  hf_detail::g_splitk_counters = static_cast<unsigned int *>(zeroed_ints);
  hf_detail::g_splitk_counter_ints = zeroed_ints ? n_ints : 0;
  return HF_OK;
}

int hf_detail::launch_splitk_reduce(ConvParams &P, bool with_epilogue, hipStream_t st) {
  const long long slab = (long long)max(1, P.groups) * P.batch * P.cout * P.out_h * P.out_w;
  long long g = (slab + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(splitk_reduce, dim3((int)g), dim3(256), 0, st, P, slab, with_epilogue ? 1 : 0);
  return hf_launch_status();
}
namespace {

// Tuning hook (hf_debug_set_dispatch): 0 = built-in heuristics.
thread_local int g_force_same = 0, g_force_up = 0;  // per-thread (hf_debug_set_dispatch)
thread_local int g_last_cfg = 0;   // tile configuration id of the last conv call (see the dispatch switches)
thread_local int g_last_path = 0;  // 1 = general kernel, 2 = pipelined kernel, 3 = split-K (general kernel + reduce)

// split-K through `workspace` with the small-plane general configuration
template <bool UP, int TAPS>
int run_splitk(ConvParams &P, int sk, float *workspace, long long workspace_floats, hipStream_t st) {
  if (!workspace || workspace_floats < (long long)sk * max(1, P.groups) * P.batch * P.cout * P.out_h * P.out_w)
    return HF_E_WORKSPACE;
  const int nchunks = (P.cin + KC - 1) / KC;
  P.chunks_per_split = (nchunks + sk - 1) / sk;
  // no EMPTY split (5 splits of 2 chunks over 8 chunks): the pipelined kernel prefetches its first chunk
  // unconditionally, an empty split would read past the end of x (a memory fault when x ends a mapped region)
  sk = (nchunks + P.chunks_per_split - 1) / P.chunks_per_split;
  P.splits = sk;
  P.partial = workspace;
  P.counters = nullptr;  // set by the launcher once the grid is known
  int rc = HF_E_INVALID;
  if (!UP && TAPS == 9) {  // double-buffered 64 co x 64 px kernel when the shape qualifies
    rc = (P.stride == 2) ? launch_conv_pipe<1, 1, 2, 2, false, 0, 2>(P, st) : launch_conv_pipe<1, 1, 2, 2, false, 0, 1>(P, st);
    if (rc == HF_E_INVALID) P.splits = sk;
  }
  if (rc == HF_E_INVALID) rc = launch_conv<1, 1, 2, 2, UP, TAPS>(P, st);
  if (rc != HF_OK) return rc;
  g_last_path = 3;
  if (P.counters) return HF_OK;  // the tiles' last blocks finished the sum and the epilogue in the kernel
  return launch_splitk_reduce(P, !UP, st);
}

// 3x3 stride-1 convolution (modulated or plain): pipelined kernel when the shape
// qualifies, else the general one.
int run_conv3x3_s1(ConvParams &P, float *workspace, long long workspace_floats, hipStream_t st) {
  const int batch = P.batch * max(1, P.groups), cout = P.cout, h = P.h, w = P.w;  // block counts scale with groups
  {
    const int sk = (g_force_same == 0) ? splitk_plan(batch, P.cin, cout, h, w) : 1;
    if (sk > 1) return run_splitk<false, 9>(P, sk, workspace, workspace_floats, st);
  }
  // Candidate tile configurations, tried in order until one accepts the shape.
  // 3x = DMA-staged (rows of 32 pixels), 1x = register-staged pipelined kernel.
  int cands[3] = {g_force_same, 0, 0};
  if (g_force_same == 0) {
    // Built-in heuristics (tools/bench_layers.py sweeps on MI355X, profiles/): the largest tile
    // that still yields >= ~1.5-2 resident blocks per CU (256 CUs).
    const long long per_img256 = ((long long)h * w + 255) / 256, per_img64 = ((long long)h * w + 63) / 64;
    const long long nb128 = batch * per_img256 * (cout / 128), nb64 = batch * per_img256 * (cout / 64);
    const long long nb32 = batch * per_img256 * (cout / 32), nb15 = batch * per_img64 * (cout / 64);
    if (cout % 128 == 0 && nb128 >= 512) { cands[0] = 31; cands[1] = 11; }
    else if (cout % 64 == 0 && nb64 >= 512) { cands[0] = 32; cands[1] = 12; }
    else if (cout % 32 == 0 && nb32 >= 256) {
      if (P.cin >= 64) { cands[0] = 33; cands[1] = 13; } else cands[0] = 13;  // few chunks: DMA prologue not amortised
    }
    else if (cout % 64 == 0 && nb15 >= 256) cands[0] = 15;
    else if (cout % 32 == 0) cands[0] = 13;
    else cands[0] = 2;
  }
  int rc = HF_E_INVALID;
  for (int ci = 0; ci < 3 && rc == HF_E_INVALID && cands[ci] != 0; ++ci) {
    const int cfg = cands[ci];
    g_last_cfg = cfg;
    switch (cfg) {
      case 11: rc = launch_conv_pipe<2, 2, 2, 4, false>(P, st); break;  // 128 co x 256 px, 8 waves
      case 12: rc = launch_conv_pipe<2, 2, 1, 4, false>(P, st); break;  //  64 co x 256 px, 4 waves
      case 13: rc = launch_conv_pipe<1, 2, 1, 4, false>(P, st); break;  //  32 co x 256 px, 4 waves
      case 14: rc = launch_conv_pipe<2, 2, 2, 2, false>(P, st); break;  // 128 co x 128 px, 4 waves
      case 15: rc = launch_conv_pipe<1, 1, 2, 2, false>(P, st); break;  //  64 co x  64 px, 4 waves
      case 16: rc = launch_conv_pipe<1, 4, 1, 4, false>(P, st); break;  //  32 co x 512 px, 4 waves
      case 31: rc = launch_conv_dma<2, 2, 2, 4>(P, st); break;          // DMA-staged, tile shapes of 11..14
      case 32: rc = launch_conv_dma<2, 2, 1, 4>(P, st); break;
      case 33: rc = launch_conv_dma<1, 2, 1, 4>(P, st); break;
      case 34: rc = launch_conv_dma<2, 2, 2, 2>(P, st); break;
      case 91: rc = launch_conv_pipe<2, 2, 2, 4, false, 1>(P, st); break;  // timing ablations of cfg 11
      case 92: rc = launch_conv_pipe<2, 2, 2, 4, false, 2>(P, st); break;
      default: break;
    }
  }
  g_last_path = 2;
  if (rc != HF_E_INVALID) return rc;
  g_last_path = 1;
  const long long pixels = (long long)batch * h * w;
  if (cout <= 32) return launch_conv<1, 2, 1, 4, false, 9>(P, st);       // 32 co x 256 px
  if (pixels <= 8192) return launch_conv<1, 1, 2, 2, false, 9>(P, st);   // 64 co x 64 px (small planes)
  if (cout <= 64) return launch_conv<2, 2, 1, 4, false, 9>(P, st);       // 64 co x 256 px
  return launch_conv<2, 2, 2, 2, false, 9>(P, st);                        // 128 co x 128 px
}

}  // namespace

extern "C" long long hf_modconv_workspace_floats(int batch, int cin, int cout, int h, int w, int upsample) {
  if (batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0) return 0;
  const int s = splitk_plan(batch, cin, cout, h, w, !upsample);
  if (s <= 1) return 0;
  const long long oh = upsample ? 2 * h + 1 : h, ow = upsample ? ((2 * w + 1 + 3) & ~3) : w;
  return (long long)s * batch * cout * oh * ow;
}

extern "C" int hf_modconv3x3_f32(float *out, const float *x, const float *wt, const float *s, const float *d,
                                 const float *noise, const float *noise_w, long long noise_bstride,
                                 const float *bias, int batch, int cin, int cout, int h, int w, float alpha,
                                 float scale, float *workspace, long long workspace_floats, void *stream) {
  if (!out || !x || !wt || batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0 || (noise && !noise_w))
    return HF_E_INVALID;
  ConvParams P{};
  P.out = out; P.x = x; P.wt = wt; P.s = s; P.d = d; P.noise = noise; P.noise_w = noise_w; P.bias = bias;
  P.s_bstride = cin; P.d_bstride = cout;
  P.groups = 1;
  P.noise_bstride = noise_bstride;
  P.batch = batch; P.cin = cin; P.cout = cout; P.h = h; P.w = w; P.out_h = h; P.out_w = w; P.out_wv = w;
  P.stride = 1;
  P.act = bias ? ACT_LRELU : ACT_NONE;
  P.alpha = alpha; P.scale = scale;
  return run_conv3x3_s1(P, workspace, workspace_floats, (hipStream_t)stream);
}

extern "C" int hf_modconv_up_pitch(int w) { return (2 * w + 1 + 3) & ~3; }

extern "C" int hf_modconv3x3_up_f32(float *tmp, const float *x, const float *wt, const float *s,
                                    const float *d, int batch, int cin, int cout, int h, int w, int tmp_pitch,
                                    float *workspace, long long workspace_floats, void *stream) {
  if (!tmp || !x || !wt || batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0 || tmp_pitch < 2 * w + 1)
    return HF_E_INVALID;
  ConvParams P{};
  P.out = tmp; P.x = x; P.wt = wt; P.s = s; P.d = d;
  P.s_bstride = cin; P.d_bstride = cout;
  P.groups = 1;
  P.batch = batch; P.cin = cin; P.cout = cout; P.h = h; P.w = w; P.out_h = 2 * h + 1; P.out_w = tmp_pitch;
  P.out_wv = 2 * w + 1;
  P.stride = 1;
  hipStream_t st = (hipStream_t)stream;
  {
    const int sk = (g_force_up == 0) ? splitk_plan(batch, cin, cout, h, w, false) : 1;
    if (sk > 1) return run_splitk<true, 9>(P, sk, workspace, workspace_floats, st);
  }
  int cfg = g_force_up;
  if (cfg == 0) {
    const long long per_img256 = ((long long)h * w + 255) / 256;
    const long long nb24 = batch * per_img256 * (cout / 64);
    if (cout % 64 == 0 && nb24 >= 256) cfg = 24;
    else if (cout % 64 == 0) cfg = 23;
    else if (cout % 32 == 0) cfg = 25;
    else cfg = 1;
  }
  g_last_cfg = cfg;
  int rc = HF_E_INVALID;
  switch (cfg) {
    case 21: rc = launch_conv_pipe<1, 2, 2, 2, true>(P, st); break;  // 64 co x 128 px x 4 phases, 4 waves
    case 22: rc = launch_conv_pipe<1, 2, 1, 4, true>(P, st); break;  // 32 co x 256 px x 4 phases, 4 waves
    case 23: rc = launch_conv_pipe<1, 1, 2, 2, true>(P, st); break;  // 64 co x  64 px x 4 phases, 4 waves
    case 24: rc = launch_conv_pipe<1, 2, 2, 4, true>(P, st); break;  // 64 co x 256 px x 4 phases, 8 waves
    case 25: rc = launch_conv_pipe<1, 1, 1, 4, true>(P, st); break;  // 32 co x 128 px x 4 phases, 4 waves
    default: break;
  }
  g_last_path = 2;
  if (rc != HF_E_INVALID) return rc;  // interior + rim families in one pipelined launch
  g_last_path = 1;
  if (cout <= 32) return launch_conv<1, 1, 1, 4, true, 9>(P, st);  // 32 co x 128 px x 4 phases
  return launch_conv<1, 1, 2, 2, true, 9>(P, st);                   // 64 co x 64 px x 4 phases
}

extern "C" long long hf_conv2d_workspace_floats(int batch, int cin, int cout, int h, int w, int k,
                                                int stride, int groups) {
  if (batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0 || stride <= 0 || groups <= 0) return 0;
  (void)k;
  const int oh = (h - 1) / stride + 1, ow = (w - 1) / stride + 1;
  const int s = splitk_plan(batch * groups, cin, cout, oh, ow);
  return s <= 1 ? 0 : (long long)s * groups * batch * cout * oh * ow;
}

extern "C" int hf_conv2d_f32(float *out, const float *x, const float *wt, const float *in_scale,
                             const float *in_shift, const float *out_scale, const float *bias, int act,
                             const float *slope, float alpha, const float *residual, int batch, int cin,
                             int cout, int h, int w, int k, int stride, int groups, long long x_group_stride,
                             float *workspace, long long workspace_floats, void *stream) {
  if (!out || !x || !wt || batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0 || (k != 1 && k != 3 && k != 7) ||
      (stride != 1 && stride != 2) || groups < 1 || x_group_stride < 0)
    return HF_E_INVALID;
  const int residual_pre = (act & HF_ACT_RESIDUAL_FIRST) ? 1 : 0;  // residual added BEFORE the activation (ResNet BasicBlock)
  act &= ~HF_ACT_RESIDUAL_FIRST;
  if (act < ACT_NONE || act > ACT_QGELU || (act == ACT_PRELU && !slope)) return HF_E_INVALID;
  if (groups > 1 && (in_scale || in_shift)) return HF_E_INVALID;  // grouped form: plain conv + epilogue
  ConvParams P{};
  P.residual_pre = residual_pre;
  P.out = out; P.x = x; P.wt = wt; P.s = in_scale; P.t = in_shift; P.d = out_scale; P.bias = bias;
  P.slope = slope; P.residual = residual;
  P.s_bstride = 0; P.d_bstride = 0;
  P.batch = batch; P.cin = cin; P.cout = cout; P.h = h; P.w = w;
  P.out_h = (h - 1) / stride + 1; P.out_w = (w - 1) / stride + 1; P.out_wv = P.out_w;
  P.stride = stride;
  P.act = act; P.alpha = alpha; P.scale = 1.0f;
  P.groups = groups; P.x_gstride = x_group_stride; P.wt_gstride = (long long)k * k * cin * cout;
  hipStream_t st = (hipStream_t)stream;
  if (k == 7) {  // 7x7 (the 3 -> 64 stride-2 stem of BiSeNet's ResNet18, resnet.py:59): general kernel, 64 co x 64 px
    if (groups > 1) return HF_E_INVALID;
    g_last_cfg = 7;
    g_last_path = 1;
    return launch_conv<1, 1, 2, 2, false, 49>(P, st);
  }
  if (k == 3 && stride == 1) return run_conv3x3_s1(P, workspace, workspace_floats, st);
  // strided 3x3 and 1x1 (a few % of the encoders' FLOPs): split-K when the grid is small,
  // the pipelined stride-2 kernel for the large 3x3 ones, else the general kernel
  const int eb = batch * groups;
  const int sk = splitk_plan(eb, cin, cout, P.out_h, P.out_w);
  g_last_cfg = 2;
  if (sk > 1) return (k == 3) ? run_splitk<false, 9>(P, sk, workspace, workspace_floats, st)
                              : run_splitk<false, 1>(P, sk, workspace, workspace_floats, st);
  const long long opix = (long long)eb * P.out_h * P.out_w;
  if (k == 3) {
    int rc = HF_E_INVALID;
    const long long nb128 = (long long)eb * ((P.out_h * P.out_w + 127) / 128) * (cout / 128);
    if (cout % 128 == 0 && nb128 >= 256) { g_last_cfg = 44; rc = launch_conv_pipe<2, 2, 2, 2, false, 0, 2>(P, st); }
    if (rc == HF_E_INVALID && cout % 64 == 0) { g_last_cfg = 45; rc = launch_conv_pipe<1, 1, 2, 2, false, 0, 2>(P, st); }
    g_last_path = 2;
    if (rc != HF_E_INVALID) return rc;
    g_last_cfg = 2;
    g_last_path = 1;
    if (cout > 64 && opix > 8192) return launch_conv<2, 2, 2, 2, false, 9>(P, st);  // 128 co x 128 px
    return launch_conv<1, 1, 2, 2, false, 9>(P, st);
  }
  g_last_path = 1;
  if (cout > 64 && opix > 8192) return launch_conv<2, 2, 2, 2, false, 1>(P, st);
  return launch_conv<1, 1, 2, 2, false, 1>(P, st);
}

namespace hf_detail {
void note_path(int path, int cfg) {
  g_last_path = path;
  g_last_cfg = cfg;
}
}  // namespace hf_detail

extern "C" int hf_debug_last_path(void) { return g_last_path * 100 + (g_last_path == 3 ? 0 : g_last_cfg); }

extern "C" int hf_debug_set_dispatch(int same_cfg, int up_cfg) {
  g_force_same = (same_cfg >= 50 && same_cfg < 60) ? 0 : same_cfg;
  g_force_h = (same_cfg >= 50 && same_cfg < 60) ? same_cfg : 0;
  g_force_up = up_cfg;
  return HF_OK;
}
