// modconv.hip - fp32 MFMA implicit-GEMM kernels for StyleGAN2's modulated 3x3 convs.
//
// hf_modconv3x3_f32     same-resolution StyledConv: ModulatedConv2d (+demod) +
//                       NoiseInjection + FusedLeakyReLU
//                       (reference models/stylegan2/model.py:238-250, :273-277, :337-343)
// hf_modconv3x3_up_f32  upsampling ModulatedConv2d, conv_transpose2d(stride 2) part
//                       (model.py:252-262); the blur + noise + activation part is
//                       hf_blur_noise_bias_act_f32 (upfirdn2d.hip).
//
// Formulation (MI355X-first, not the reference's per-sample grouped conv):
//   y[b,co,p] = d[b,co] * sum_{tap,ci} wt[tap,ci,co] * (s[b,ci] * x[b,ci,p+tap])
// i.e. the modulation s is applied to the ACTIVATIONS while they are staged
// into LDS and the demodulation d to the OUTPUTS in the epilogue, so one weight
// tensor is shared by the whole batch: an implicit GEMM with M = cout,
// N = batch*H*W pixels, K = 9*cin, run on v_mfma_f32_32x32x2_f32 (exact fp32,
// bitwise an fmaf chain; 157 TFLOP/s peak = the roofline of these layers).
//
// Transposed conv without wasted zero-insert flops: output (2Y+pr, 2X+pc) only
// receives taps with ky = pr, kx = pc (mod 2), so each of the 9 taps feeds one
// of 4 phase accumulators: same MFMA count as a 3x3 conv at INPUT resolution.
//
// Work decomposition: block = 256 threads = 4 waves (64 lanes each); block tile
// = CT output channels x PT pixels.  Pixels of a tile are (image, row, col)
// triples described at run time by a TileGeom, so one kernel serves 1024^2
// planes (8x32 tiles), 4x4 planes (8 whole images per tile) and the 1-pixel
// wide rim tiles of the (2H+1)x(2W+1) transposed-conv output.
// K loop: chunks of KC=8 input channels; per chunk the block stages
// wt[9][KC][CT] and the modulated halo tile x[KC][images][rows+halo][cols+halo]
// into LDS (coalesced row segments, 16 B weight loads), then every wave issues
// 9*KC/2 MFMAs per (co tile, pixel group) reading one fp32 A and B operand per
// lane with conflict-free ds_read_b32 (A: 32 consecutive co, B: 32 consecutive
// pixels; lanes 32-63 take the next input channel).
#include "hf_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KC = 8;            // input channels per LDS stage (4 MFMA k-steps)
constexpr int kThreads = 256;    // 4 waves
constexpr int kMaxElemPerCi = 4; // halo-tile elements per thread per channel

struct TileGeom {
  int y0, x0;      // origin of this tile family in the pixel domain
  int dh, dw;      // extent of the family (pixels)
  int lg_tw, lg_th;
  int lg_nb;       // images per tile = 1 << lg_nb ; (nb*th*tw == PT)
  int tiles_x, tiles_y, tiles_b;
  int first_block; // linear block id of the family's first tile
};

struct ConvParams {
  float *out;
  const float *x, *wt, *s, *d, *noise, *noise_w, *bias;
  long long noise_bstride;
  int batch, cin, cout, h, w;  // input plane h x w
  int out_h, out_w;            // output plane (same-res: h,w ; up: 2h+1, 2w+1)
  float alpha, scale;
  int n_geom;
  int xs_max;                  // LDS floats reserved per staged channel
  TileGeom g[3];
};

// CT_TILES x PG 32x32 MFMA tiles per wave; WAVES_CO x WAVES_PX = 4 waves.
template <int CT_TILES, int PG, int WAVES_CO, int WAVES_PX, bool UP>
__global__ __launch_bounds__(kThreads) void modconv_mfma(const ConvParams P) {
  static_assert(WAVES_CO * WAVES_PX == 4, "4 waves per block");
  constexpr int CT = 32 * CT_TILES * WAVES_CO;
  constexpr int PT = 32 * PG * WAVES_PX;
  constexpr int NPH = UP ? 4 : 1;
  constexpr int HALO = UP ? 1 : 2;  // UP needs x[Y-1], x[X-1] only; same-res needs +-1

  HF_DYN_LDS;
  float *wl = reinterpret_cast<float *>(hf_dyn_lds);  // [9][KC][CT]
  float *xl = wl + 9 * KC * CT;                       // [KC][xs_max]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int li = lane & 31;  // MFMA row (A) / column (B) index
  const int lh = lane >> 5;  // MFMA k index within the k=2 step
  const int wave_co = (wave / WAVES_PX) * (32 * CT_TILES);
  const int wave_pg = (wave % WAVES_PX) * PG;
  const int co0 = blockIdx.y * CT;

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
  const int wp = tw + HALO, hp = th + HALO;
  const int xs = nb * hp * wp;         // LDS floats per staged channel (<= xs_max)
  const int ty0 = G.y0 + ty * th;      // first pixel row / col of the tile
  const int tx0 = G.x0 + tx * tw;
  const int b0 = tb * nb;
  const long long plane = (long long)P.h * P.w;
  const float *xb = P.x + (long long)b0 * P.cin * plane;

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
      const int ys = ty0 + hy - 1, xc = tx0 + hx - 1;
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
    pixoff[g] = im * hp * wp + py * wp + px;
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

  for (int ci0 = 0; ci0 < P.cin; ci0 += KC) {
    __syncthreads();  // previous chunk fully consumed

    // ---- stage weights: wl[tap][kc][c] = wt[tap][ci0+kc][co0+c] ---------------
    if (cout_vec4) {
      constexpr int C4 = CT / 4;
      for (int i = tid; i < 9 * KC * C4; i += kThreads) {
        const int c4 = i % C4;
        const int kc = (i / C4) % KC;
        const int tap = i / (C4 * KC);
        const int ci = ci0 + kc, co = co0 + c4 * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ci < P.cin && co < P.cout)
          v = *reinterpret_cast<const float4 *>(P.wt + ((long long)tap * P.cin + ci) * P.cout + co);
        *reinterpret_cast<float4 *>(wl + (tap * KC + kc) * CT + c4 * 4) = v;
      }
    } else {
      for (int i = tid; i < 9 * KC * CT; i += kThreads) {
        const int c = i % CT;
        const int kc = (i / CT) % KC;
        const int tap = i / (CT * KC);
        const int ci = ci0 + kc, co = co0 + c;
        float v = 0.f;
        if (ci < P.cin && co < P.cout) v = P.wt[((long long)tap * P.cin + ci) * P.cout + co];
        wl[(tap * KC + kc) * CT + c] = v;
      }
    }

    // ---- stage the modulated halo tile: xl[kc][idx] = s[b,ci] * x[b,ci,ys,xc] ---
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
            if (P.s) v *= P.s[(long long)(b0 + e_img[e]) * P.cin + ci];
          }
          xl[kc * P.xs_max + idx] = v;
        }
      }
    }
    __syncthreads();

    // ---- MFMA over the chunk: 9 taps x KC/2 k-steps -----------------------------
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap % 3;
      // LDS offset of the tap's source pixel relative to pixoff
      const int toff = UP ? ((ky == 2 ? 0 : 1) * wp + (kx == 2 ? 0 : 1)) : (ky * wp + kx);
      const int ph = UP ? ((ky & 1) * 2 + (kx & 1)) : 0;
#pragma unroll
      for (int kk = 0; kk < KC / 2; ++kk) {
        float a[CT_TILES], bq[PG];
#pragma unroll
        for (int ct = 0; ct < CT_TILES; ++ct) a[ct] = a_base[(tap * KC + 2 * kk) * CT + ct * 32];
#pragma unroll
        for (int g = 0; g < PG; ++g) bq[g] = b_base[2 * kk * P.xs_max + pixoff[g] + toff];
#pragma unroll
        for (int ct = 0; ct < CT_TILES; ++ct)
#pragma unroll
          for (int g = 0; g < PG; ++g)
            acc[ph][ct][g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ct], bq[g], acc[ph][ct][g], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: D[row = co, col = pixel]; row = (r&3) + 8*(r>>2) + 4*lh, col = li ----
  const long long oplane = (long long)P.out_h * P.out_w;
  const float nw = (!UP && P.noise) ? P.noise_w[0] : 0.0f;
#pragma unroll
  for (int g = 0; g < PG; ++g) {
    const int p = (wave_pg + g) * 32 + li;
    const int px = p & (tw - 1);
    const int py = (p >> G.lg_tw) & (th - 1);
    const int im = p >> (G.lg_tw + G.lg_th);
    const int Y = ty0 + py, X = tx0 + px, b = b0 + im;
    const bool pv = (Y < G.y0 + G.dh) && (X < G.x0 + G.dw) && (b < P.batch);
    if (!pv) continue;
    float nz = 0.0f;
    if (!UP && P.noise) nz = nw * P.noise[(long long)b * P.noise_bstride + (long long)Y * P.w + X];
#pragma unroll
    for (int ct = 0; ct < CT_TILES; ++ct) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wave_co + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (co >= P.cout) continue;
        const float dm = P.d ? P.d[(long long)b * P.cout + co] : 1.0f;
        float *ob = P.out + ((long long)b * P.cout + co) * oplane;
        if (UP) {
#pragma unroll
          for (int ph = 0; ph < 4; ++ph) {
            const int ro = 2 * Y + (ph >> 1), cc = 2 * X + (ph & 1);
            if (ro < P.out_h && cc < P.out_w) ob[(long long)ro * P.out_w + cc] = acc[ph][ct][g][r] * dm;
          }
        } else {
          float v = acc[0][ct][g][r] * dm + nz;
          if (P.bias) v = hf_lrelu(v + P.bias[co], P.alpha, P.scale);
          ob[(long long)Y * P.out_w + X] = v;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------
// host side: tile-geometry selection
// ------------------------------------------------------------------------------

inline int ilog2(int v) {
  int l = 0;
  while ((1 << (l + 1)) <= v) ++l;
  return l;
}
inline int pow2_floor(int v) { return 1 << ilog2(v); }
inline int pow2_ceil(int v) { return (v & (v - 1)) ? (pow2_floor(v) << 1) : v; }

// Tiles of `pt` pixels over a dh x dw domain (per image) of `batch` images.
// Prefers full 32-pixel rows; small planes put several images in one tile.
inline TileGeom make_geom(int y0, int x0, int dh, int dw, int batch, int pt, int first_block) {
  TileGeom g;
  g.y0 = y0; g.x0 = x0; g.dh = dh; g.dw = dw;
  int tw = pow2_ceil(dw);
  if (tw > 32 && dh > 1) tw = 32;  // rows of 32 consecutive pixels when the domain is 2-D
  if (tw > pt) tw = pt;
  int th = pow2_ceil(dh);
  if (th > pt / tw) th = pt / tw;
  int nb = pt / (tw * th);
  if (nb > pow2_ceil(batch)) nb = pow2_ceil(batch);  // never stage images that do not exist
  g.lg_tw = ilog2(tw); g.lg_th = ilog2(th); g.lg_nb = ilog2(nb);
  g.tiles_x = hf_cdiv(dw, tw);
  g.tiles_y = hf_cdiv(dh, th);
  g.tiles_b = hf_cdiv(batch, nb);
  g.first_block = first_block;
  return g;
}
inline int geom_blocks(const TileGeom &g) { return g.tiles_x * g.tiles_y * g.tiles_b; }
inline int geom_xs(const TileGeom &g, int halo) {
  return (1 << g.lg_nb) * ((1 << g.lg_th) + halo) * ((1 << g.lg_tw) + halo);
}

template <int CT_TILES, int PG, int WAVES_CO, int WAVES_PX, bool UP>
int launch_modconv(ConvParams &P, hipStream_t st) {
  constexpr int CT = 32 * CT_TILES * WAVES_CO;
  constexpr int PT = 32 * PG * WAVES_PX;
  constexpr int HALO = UP ? 1 : 2;
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
    P.g[0] = make_geom(0, 0, P.h, P.w, P.batch, PT, 0);
    nblocks = geom_blocks(P.g[0]);
  }
  int xs_max = 0;
  for (int i = 0; i < P.n_geom; ++i) xs_max = max(xs_max, geom_xs(P.g[i], HALO));
  if (xs_max > kMaxElemPerCi * kThreads) return HF_E_INVALID;
  for (int i = 0; i < P.n_geom; ++i)  // staging offsets are 32-bit, relative to the tile's first image
    if (((long long)P.cin << P.g[i].lg_nb) * P.h * P.w >= (1LL << 31)) return HF_E_INVALID;
  P.xs_max = (xs_max + 3) & ~3;
  const size_t lds = (size_t)(9 * KC * CT + KC * P.xs_max) * sizeof(float);
  dim3 grid(nblocks, hf_cdiv(P.cout, CT));
  if (grid.y > 65535) return HF_E_INVALID;
  hipLaunchKernelGGL((modconv_mfma<CT_TILES, PG, WAVES_CO, WAVES_PX, UP>), grid, dim3(kThreads), lds, st, P);
  return hf_launch_status();
}

}  // namespace

extern "C" int hf_modconv3x3_f32(float *out, const float *x, const float *wt, const float *s, const float *d,
                                 const float *noise, const float *noise_w, long long noise_bstride,
                                 const float *bias, int batch, int cin, int cout, int h, int w, float alpha,
                                 float scale, void *stream) {
  if (!out || !x || !wt || batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0 || (noise && !noise_w))
    return HF_E_INVALID;
  ConvParams P{};
  P.out = out; P.x = x; P.wt = wt; P.s = s; P.d = d; P.noise = noise; P.noise_w = noise_w; P.bias = bias;
  P.noise_bstride = noise_bstride;
  P.batch = batch; P.cin = cin; P.cout = cout; P.h = h; P.w = w; P.out_h = h; P.out_w = w;
  P.alpha = alpha; P.scale = scale;
  hipStream_t st = (hipStream_t)stream;
  const long long pixels = (long long)batch * h * w;
  if (cout <= 32) return launch_modconv<1, 2, 1, 4, false>(P, st);       // 32 co x 256 px
  if (pixels <= 8192 || cout <= 64) {
    if (pixels <= 8192) return launch_modconv<1, 1, 2, 2, false>(P, st); // 64 co x 64 px (small planes)
    return launch_modconv<2, 2, 1, 4, false>(P, st);                      // 64 co x 256 px
  }
  return launch_modconv<2, 2, 2, 2, false>(P, st);                        // 128 co x 128 px
}

extern "C" int hf_modconv3x3_up_f32(float *tmp, const float *x, const float *wt, const float *s,
                                    const float *d, int batch, int cin, int cout, int h, int w,
                                    void *stream) {
  if (!tmp || !x || !wt || batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0) return HF_E_INVALID;
  ConvParams P{};
  P.out = tmp; P.x = x; P.wt = wt; P.s = s; P.d = d;
  P.batch = batch; P.cin = cin; P.cout = cout; P.h = h; P.w = w; P.out_h = 2 * h + 1; P.out_w = 2 * w + 1;
  hipStream_t st = (hipStream_t)stream;
  if (cout <= 32) return launch_modconv<1, 1, 1, 4, true>(P, st);  // 32 co x 128 px x 4 phases
  return launch_modconv<1, 1, 2, 2, true>(P, st);                   // 64 co x 64 px x 4 phases
}
