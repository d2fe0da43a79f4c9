// conv_common.h - declarations shared by the convolution kernels (modconv.hip: exact-fp32
// MFMA; convh.hip: fp16-split MFMA): launch parameters, tile geometry, the epilogue.
#pragma once
#include "hf_common.h"

#ifndef HF_STORE_OUT
#ifdef HF_NT_STORES
#define HF_STORE_OUT(p, v) __builtin_nontemporal_store((v), (p))
#else
#define HF_STORE_OUT(p, v) (*(p) = (v))
#endif
#endif

namespace hf_detail {

typedef float f32x16 __attribute__((ext_vector_type(16)));
struct __attribute__((packed, aligned(4))) f32x2u {  // 8-byte value with 4-byte alignment
  float x, y;
};

constexpr int KC = 8;             // input channels per LDS stage (4 MFMA k-steps)
constexpr int kThreads = 256;     // general kernel: 4 waves
constexpr int kMaxElemPerCi = 4;  // general kernel: halo-tile elements per thread per channel

enum { ACT_NONE = 0, ACT_LRELU = 1, ACT_PRELU = 2, ACT_QGELU = 3 };  // QGELU: x * sigmoid(1.702 x) (CLIP's QuickGELU)

struct TileGeom {
  int y0, x0;      // origin of this tile family in the OUTPUT pixel domain
  int dh, dw;      // extent of the family (pixels)
  int lg_tw, lg_th;
  int lg_nb;       // images per tile = 1 << lg_nb ; (nb*th*tw == PT)
  int tiles_x, tiles_y, tiles_b;
  int first_block; // linear block id of the family's first tile
};

struct ConvParams {
  float *out;
  const float *x, *wt;
  const float *s;        // input scale  s[b*s_bstride + ci]   (modulation / pre-conv BN scale) or null
  const float *t;        // input shift  t[ci]                 (pre-conv BN shift) or null
  const float *d;        // output scale d[b*d_bstride + co]   (demodulation / post-conv BN scale) or null
  const float *noise, *noise_w, *bias;
  const float *slope;    // PReLU slopes [cout]
  const float *residual; // added after the activation (residual_pre: before it), same shape as out
  int residual_pre;
  long long noise_bstride;
  int s_bstride, d_bstride;
  int batch, cin, cout, h, w;  // input plane h x w
  int out_h, out_w;            // output plane (stride-1: h,w ; stride-2: ceil(h/2) ; up: 2h+1 rows of pitch out_w)
  int out_wv;                  // up: valid columns (2w+1) of the out_w-pitched rows; otherwise == out_w
  int stride;                  // 1 or 2 (general kernel only)
  int act;
  float alpha, scale;
  int n_geom;
  int xs_max;                  // LDS floats reserved per staged channel
  int groups;                  // grouped launch: blockIdx.y = group * co_tiles + co_tile (0/1 = off)
  int co_tiles;                // cout tiles per group
  long long x_gstride;         // floats between the inputs of consecutive groups (0 = shared input)
  long long wt_gstride;        // taps*cin*cout
  long long zslab;             // split-K: floats per z slab of `partial` (= groups*batch*cout*oh*ow)
  int splits;                  // split-K: blockIdx.z handles chunks [z*cps, (z+1)*cps); 1 = off
  int chunks_per_split;
  int swap_xy;                 // convh_enc.hip: the grid is (columns, tiles) instead of (tiles, columns) - blocks are dispatched
                               // x-fastest, so consecutive blocks then share an INPUT tile and walk the (group, channel-tile)
                               // columns: the input tile stays in L2 while the weights stream from the Infinity Cache
                               // (run_enc picks the order with the smaller beyond-L2 traffic)
  int persist;                 // convh_enc.hip (conv_enc_h): the launch is `persist` resident blocks that walk the (lin_x x lin_y) block
  int lin_x, lin_y;            // grid of the plain form in dispatch order, block b taking b, b + persist, ... (0 = one block per tile)
  int vsplit;                  // "virtual" split-K (convh_enc.hip, gemm_h.hip; batch-invariant plans): the K partition into `splits`
                               // slabs is kept - it decides the bits - but ONE block walks all slabs, adding each slab's sum to a
                               // second accumulator set in z order (exactly what splitk_reduce adds), and runs the epilogue itself:
                               // no partial slabs in memory, no second launch.  Taken when the launch fills the chip without a
                               // real K split (a batched pass); a batch-1 launch of the same layer splits for real - same bits.
  float *partial;              // splits > 1: raw accumulators go to partial[z][b][co][oh][ow]
  unsigned int *counters;      // splits > 1, non-null: one arrival counter per (blockIdx.x, blockIdx.y) output tile, zero on entry - the
                               // LAST block of a tile to arrive adds the z slabs in z order and runs the epilogue (no second launch)
  // convh.hip, fused ToRGB: raw 1x1 modulated conv of the epilogue's output values,
  // rgb_out[b][c][Y][X] = sum_co rgb_w[co*3+c] * rgb_s[b*cout+co] * y[b][co][Y][X]  (null = off)
  const float *rgb_w, *rgb_s;
  float *rgb_out;
  // convrow.hip only (null = off): rgb_out receives the FINISHED ToRGB image instead of the raw 1x1 product -
  // + rgb_bias[c] + the x2-upsampled skip (rgb_skip [B][3][H/2][W/2], rgb_k4 = its 4x4 kernel): hf_torgb_f32's arithmetic
  const float *rgb_skip, *rgb_bias, *rgb_k4;
  // convh.hip, pre-split input: s*x already split into fp16 (hi, lo) and K-blocked
  // [batch][cin/8][h][w][8] by the producer (hf_blur_noise_bias_act_split_f16); x and s unused
  const void *xh, *xl;
  // convh.hip, split output (same-resolution epilogue): s_next[b][co] * out also written as fp16
  // (hi, lo), K-blocked [batch][cout/8][h][w][8], for a consumer that takes pre-split input
  void *oh, *ol;
  const float *s_next;
  // convh_enc.hip / gemm_h.hip split output (store_tile_rows): a_next[co] * out + t_next[co] (the consumer's input
  // affine, NULL = identity) as fp16 (hi, lo) in the K-blocked layout; P.out may then be NULL
  const float *a_next, *t_next;
  float blur_kx[4], blur_ky[4];  // convh.hip FUSE: flipped 1-D factors of the (rank-1) 4x4 blur kernel applied in the epilogue
  int rgb_slabs;               // convh.hip fused ToRGB: slabs of the raw tensor the caller allocated ([B][slabs*3][H][W])
  int dma_early;               // convh.hip PRE: issue a stage's DMAs in its first tap-step (short K loops) instead of spread
  int n_tiles;                 // convh.hip: tiles over all families; a block walks blockIdx.x + k*gridDim.x
  TileGeom g[3];
};

// Grouped launch: G independent convolutions of identical shape in one grid (the style
// heads of the e4e encoder).  Weights / per-channel vectors / outputs of group g follow
// those of group g-1; the input is shared (x_gstride 0) or per group.  Offsets of this
// block's group (all zero for ordinary launches); the kernel-argument struct itself is
// never copied (a modified copy would live in scratch memory).
struct GroupOfs {
  long long x, wt, o;  // element offsets into x, wt and out / partial / residual
  int c;               // offset into bias / slope / d
  int co_tile;
};
__device__ __forceinline__ GroupOfs group_offsets(const ConvParams &P, int bx, int by) {
  const int col = P.swap_xy ? bx : by;  // (group, channel tile) column of the block
  GroupOfs go{0, 0, 0, 0, col};
  if (P.groups > 1) {
    const int g = col / P.co_tiles;
    go.co_tile = col - g * P.co_tiles;
    go.x = (long long)g * P.x_gstride;
    go.wt = (long long)g * P.wt_gstride;
    go.o = (long long)g * P.batch * P.cout * P.out_h * P.out_w;
    go.c = g * P.cout;  // per-channel vectors (d_bstride == 0 in grouped launches)
  }
  return go;
}
__device__ __forceinline__ GroupOfs group_offsets(const ConvParams &P) { return group_offsets(P, (int)blockIdx.x, (int)blockIdx.y); }

// (`#pragma clang fp contract(on)` at the head of the two bodies below: clang applies FP contraction lexically, so a
// pragma in the CALLER does not reach these inlined bodies - without it hipcc's default `fast` contraction may fuse
// `v * d + bias` differently in the unrolled and in the remainder iterations of a grid-stride loop, which is the
// batch-size-dependent rounding the batch-invariant plans exclude.)
__device__ __forceinline__ float apply_act(float v, int act, float alpha, float scale, float slope) {
#pragma clang fp contract(on)
  if (act == ACT_LRELU) return hf_lrelu(v, alpha, scale);
  if (act == ACT_PRELU) return v > 0.0f ? v : v * slope;
  if (act == ACT_QGELU) return v / (1.0f + expf(-1.702f * v));
  return v;
}

// ---- split-K: the second half ---------------------------------------------------------------------------------
// out[i] = epilogue( d * sum_z partial[z][i] ): the z slabs are added in z order from 0.0f, then the per-element tail in
// this operation order - shared by the splitk_reduce kernel (modconv.hip) and by the in-kernel form below, so that the
// two give identical bits.  i = element index in out / a slab, b = image, gc = group*cout + channel, pix = Y*out_w + X.
__device__ __forceinline__ float splitk_finish(const ConvParams &P, long long i, long long b, long long gc, long long pix,
                                               float nw, bool with_epilogue) {
#pragma clang fp contract(on)
  float v = 0.0f;
  for (int z = 0; z < P.splits; ++z) v += P.partial[(long long)z * P.zslab + i];
  if (with_epilogue && P.d_bstride == 0 && !P.noise) {
    // encoder-type launches (per-channel out scale, no noise): the arithmetic of store_tile_rows - ONE fma, then the
    // activation - so that a layer gives the same bits whether its K split is real (this pass) or virtual (ConvParams::vsplit:
    // the block's own store_tile_rows epilogue)
    v = fmaf(v, P.d ? P.d[gc] : 1.0f, P.bias ? P.bias[gc] : 0.0f);
    if (P.residual && P.residual_pre) v += P.residual[i];
    const float neg = P.act == ACT_PRELU ? P.slope[gc] : (P.act == ACT_LRELU ? P.alpha : 1.0f);
    v = (v > 0.0f ? v : v * neg) * (P.act == ACT_LRELU ? P.scale : 1.0f);
    if (P.act == ACT_QGELU) v = v / (1.0f + expf(-1.702f * v));
    if (P.residual && !P.residual_pre) v += P.residual[i];
    return v;
  }
  if (P.d) v *= P.d[b * P.d_bstride + gc];
  if (with_epilogue) {
    if (P.noise) v = fmaf(nw, P.noise[b * P.noise_bstride + pix], v);
    if (P.bias) v += P.bias[gc];
    if (P.residual && P.residual_pre) v += P.residual[i];
    v = apply_act(v, P.act, P.alpha, P.scale, P.act == ACT_PRELU ? P.slope[gc] : 0.0f);
    if (P.residual && !P.residual_pre) v += P.residual[i];
  }
  return v;
}

// In-kernel second half (P.counters != NULL): every block of a split-K launch stores its raw sums into its z slab, then
// announces itself on its output tile's counter; the block that arrives LAST (all other slabs of the tile are complete and,
// after the fences, visible - agent scope: the blocks of a tile may sit on different XCDs) reads the tile's elements back
// from ALL slabs in z order and writes the finished values.  Deterministic: the order of the additions does not depend on
// which block is last.  The counter is left at zero for the next launch.  `flag`: one LDS word no wave still reads.
// Returns true in every thread of the last block.
__device__ __forceinline__ bool splitk_arrive_last(const ConvParams &P, int *flag) {
  __threadfence();   // release: this thread's slab stores before the arrival
  __syncthreads();
  if (threadIdx.x == 0) {
    const int tile = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x;
    const unsigned int old = atomicAdd(P.counters + tile, 1u);
    const int last = old == (unsigned int)(P.splits - 1);
    if (last) P.counters[tile] = 0u;
    *flag = last;
  }
  __syncthreads();
  const bool last = *flag != 0;
  if (last) __threadfence();  // acquire: the other blocks' slabs
  return last;
}

// Epilogue.  MFMA D layout: row (= co) = (r&3) + 8*(r>>2) + 4*(lane>>5), col (= pixel) = lane&31.
// Per pixel group the 16*CT_TILES output-scale / bias / slope values of the lane's output
// channels are fetched with independent loads up front (no load->wait->store chains).
//   v = acc*d + noise_w*noise + bias ; v = act(v) ; v += residual
// REDUCE: the last block of a split-K tile (splitk_arrive_last) - the same element walk, values = splitk_finish of the slabs
template <int CT_TILES, int PG, bool UP, int NPH = (UP ? 4 : 1), bool REDUCE = false>
__device__ __forceinline__ void store_tile(const ConvParams &P, const TileGeom &G, const GroupOfs &go,
                                           f32x16 (&acc)[NPH][CT_TILES][PG], int co_wave, int wave_pg, int li,
                                           int lh, int ty0, int tx0, int b0) {
  static_assert(!(REDUCE && UP), "in-kernel split-K reduction: same-resolution / strided forms only");
  const int tw = 1 << G.lg_tw, th = 1 << G.lg_th;
  const long long oplane = (long long)P.out_h * P.out_w;
  if constexpr (REDUCE) {
    const float nw_r = P.noise ? P.noise_w[0] : 0.0f;
#pragma unroll
    for (int g = 0; g < PG; ++g) {
      const int p = (wave_pg + g) * 32 + li;
      const int px = p & (tw - 1), py = (p >> G.lg_tw) & (th - 1), im = p >> (G.lg_tw + G.lg_th);
      const int Y = ty0 + py, X = tx0 + px, b = b0 + im;
      if (!((Y < G.y0 + G.dh) && (X < G.x0 + G.dw) && (b < P.batch))) continue;
      const long long pix = (long long)Y * P.out_w + X;
#pragma unroll
      for (int ct = 0; ct < CT_TILES; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co_wave + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (co >= P.cout) continue;
          const long long i = go.o + ((long long)b * P.cout + co) * oplane + pix;
          P.out[i] = splitk_finish(P, i, b, go.c + co, pix, nw_r, true);
        }
    }
    return;
  }
  const bool partial = P.splits > 1 && !P.vsplit;  // split-K: raw sums, epilogue runs in splitk_reduce (virtual split: the block holds the finished sums)
  const bool full = !UP && !partial;
  const float nw = (full && P.noise) ? P.noise_w[0] : 0.0f;
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
    if (full && P.noise) nz = nw * P.noise[(long long)b * P.noise_bstride + (long long)Y * P.out_w + X];
    float *obase = (partial ? P.partial + (long long)blockIdx.z * P.zslab : P.out) + go.o;
#pragma unroll
    for (int ct = 0; ct < CT_TILES; ++ct) {
      float dmv[16], bsv[16], slv[16];  // one co tile at a time: 48 live registers, not 48*CT_TILES
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co_wave + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int cc = min(co, P.cout - 1);
        dmv[r] = (P.d && !partial) ? P.d[(long long)b * P.d_bstride + go.c + cc] : 1.0f;
        bsv[r] = (full && P.bias) ? P.bias[go.c + cc] : 0.0f;
        slv[r] = (full && P.act == ACT_PRELU) ? P.slope[go.c + cc] : 0.0f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co_wave + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (co >= P.cout) continue;
        const long long obofs = ((long long)b * P.cout + co) * oplane;
        float *ob = obase + obofs;
        const float dm = dmv[r];
        if (UP) {
          // phases (pr,0) and (pr,1) are neighbouring columns: one 8-byte store per row
          // (rows of the (2h+1)x(2w+1) plane are only 4-byte aligned: unaligned-dword store)
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            const int ro = 2 * Y + pr, cc = 2 * X;
            if (ro >= P.out_h) continue;
            float *q = ob + (long long)ro * P.out_w + cc;
            const float v0 = acc[2 * pr][ct][g][r] * dm, v1 = acc[2 * pr + 1][ct][g][r] * dm;
            if (cc + 1 < P.out_wv) {
              f32x2u pair;
              pair.x = v0;
              pair.y = v1;
              *reinterpret_cast<f32x2u *>(q) = pair;
            } else {
              q[0] = v0;
            }
          }
        } else {
          const long long pofs = (long long)Y * P.out_w + X;
          float v = acc[0][ct][g][r] * dm;
          if (full) {
            v = v + nz + bsv[r];
            if (P.residual && P.residual_pre) v += P.residual[go.o + obofs + pofs];
            v = apply_act(v, P.act, P.alpha, P.scale, slv[r]);
            if (P.residual && !P.residual_pre) v += P.residual[go.o + obofs + pofs];
          }
          HF_STORE_OUT(&ob[pofs], v);
        }
      }
    }
  }
}

// store_tile for whole channel tiles (cout % 32 == 0, no noise, !UP): the encoder-type epilogue
//   v = act( acc * d[co] + bias[co] (+ residual) ) (+ residual),  act in {none, leaky ReLU * scale, PReLU}
// without per-element switches: the per-channel vectors arrive as 16-byte loads of the lane's 4 consecutive channels,
// the loads of a 32-channel tile (parameters, residual) are issued before its first store (a load behind a store
// waits for the store's acknowledgement), the activation is one select + multiply (negative slope 1 = identity),
// and the run-time options are tested per tile, not per element (store_tile spends ~100 cycles per element on them: as long as the
// whole K loop of a 64-channel layer).  split-K launches store the raw sums.
// MODE (round 5): the per-launch switches as compile-time constants - bit 0 = an fp32 output, bit 1 = a split output, bit 2 =
// QuickGELU; -1 = read from P (the residual variants).  With run-time switches every ELEMENT paid three taken scalar branches
// (over the QuickGELU code, around its store, around the split) and a 64-bit multiply for its address: 12-13 k cycles per block of
// the 512-pixel form whatever the layer - 1.6 K stages, a quarter of a 64-channel layer's blocks (tools/probes/trace_enc_layer.py).
// The channel rows of a pixel are now walked by pointer increments.  Same arithmetic, same bits.
template <int CT_TILES, int PG, bool RES, int MODE = -1>
__device__ __forceinline__ void store_tile_rows_impl(const ConvParams &P, const TileGeom &G, const GroupOfs &go,
                                                     f32x16 (&acc)[1][CT_TILES][PG], int co_wave, int wave_pg, int li, int lh,
                                                     int ty0, int tx0, int b0) {
  const int tw = 1 << G.lg_tw, th = 1 << G.lg_th;
  const long long oplane = (long long)P.out_h * P.out_w;
  const bool partial = P.splits > 1 && !P.vsplit;
  const bool prelu = !partial && P.act == ACT_PRELU, lrelu = !partial && P.act == ACT_LRELU;
  const bool qgelu = MODE >= 0 ? (MODE & 4) != 0 : (!partial && P.act == ACT_QGELU);
  const float neg_u = lrelu ? P.alpha : 1.0f, sc = lrelu ? P.scale : 1.0f;
  const bool res_pre = RES && P.residual_pre;
  long long pofs[PG], uofs[PG];  // fp32 element / split 16-byte unit of (image, channel 0, Y, X)
  bool pvs[PG];
#pragma unroll
  for (int g = 0; g < PG; ++g) {
    const int p = (wave_pg + g) * 32 + li;
    const int px = p & (tw - 1), py = (p >> G.lg_tw) & (th - 1), im = p >> (G.lg_tw + G.lg_th);
    const int Y = ty0 + py, X = tx0 + px, b = b0 + im;
    pvs[g] = (Y < G.y0 + G.dh) && (X < G.x0 + G.dw) && (b < P.batch);
    pofs[g] = (long long)b * P.cout * oplane + (long long)Y * P.out_w + X;
    uofs[g] = (long long)b * (P.cout >> 3) * oplane + (long long)Y * P.out_w + X;
  }
  float *obase = (partial ? P.partial + (long long)blockIdx.z * P.zslab : P.out) + go.o;
  // split output (launcher: groups == 1, no split-K): unit (b, channel block, Y, X) of 16 bytes, the lane's 4 channels = its half lh
  const bool split = MODE >= 0 ? (MODE & 2) != 0 : (!partial && P.oh != nullptr);
  const bool store32 = MODE >= 0 ? (MODE & 1) != 0 : (partial || P.out != nullptr);
  bool ovf_tile = false;
  (void)ovf_tile;
  // one 32-channel tile at a time (register budget: some callers run two blocks per CU): its per-channel vectors
  // and residual values are loaded, then its PG * 16 values stored
#pragma unroll
  for (int ct = 0; ct < CT_TILES; ++ct) {
    float4 dm[4], bs[4], sl[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c4 = go.c + co_wave + ct * 32 + 8 * q + 4 * lh;
      dm[q] = (P.d && !partial) ? *reinterpret_cast<const float4 *>(P.d + c4) : make_float4(1.0f, 1.0f, 1.0f, 1.0f);
      bs[q] = (P.bias && !partial) ? *reinterpret_cast<const float4 *>(P.bias + c4) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      sl[q] = prelu ? *reinterpret_cast<const float4 *>(P.slope + c4) : make_float4(neg_u, neg_u, neg_u, neg_u);
    }
    float rv[RES ? PG : 1][16];
    if (RES) {
#pragma unroll
      for (int g = 0; g < PG; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co_wave + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          rv[RES ? g : 0][r] = pvs[g] ? P.residual[go.o + pofs[g] + (long long)co * oplane] : 0.0f;
        }
    }
#pragma unroll
    for (int g = 0; g < PG; ++g) {
      if (!pvs[g]) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float dmv[4] = {dm[q].x, dm[q].y, dm[q].z, dm[q].w};
        const float bsv[4] = {bs[q].x, bs[q].y, bs[q].z, bs[q].w};
        const float slv[4] = {sl[q].x, sl[q].y, sl[q].z, sl[q].w};
        float *ob = obase + pofs[g] + (long long)(co_wave + ct * 32 + 8 * q + 4 * lh) * oplane;
        float vq[4];
#pragma unroll
        for (int k = 0; k < 4; ++k, ob += oplane) {
          const int r = 4 * q + k;
          float v = fmaf(acc[0][ct][g][r], dmv[k], bsv[k]);
          if (RES && res_pre) v += rv[RES ? g : 0][r];
          v = (v > 0.0f ? v : v * slv[k]) * sc;
          if (qgelu) v = v / (1.0f + expf(-1.702f * v));  // uniform branch (per launch)
          if (RES && !res_pre) v += rv[RES ? g : 0][r];
          if (store32) HF_STORE_OUT(ob, v);
          vq[k] = v;
        }
#ifdef HF_WANT_F16_SPLIT
        if (split) {
          const int c4 = go.c + co_wave + ct * 32 + 8 * q + 4 * lh;
          float vs[4] = {vq[0], vq[1], vq[2], vq[3]};
          if (P.a_next || P.t_next) {  // one fma like hf_split_activation_f16's (bit-equal to the two-pass form)
            const float4 an = P.a_next ? *reinterpret_cast<const float4 *>(P.a_next + c4) : make_float4(1.0f, 1.0f, 1.0f, 1.0f);
            const float4 tn = P.t_next ? *reinterpret_cast<const float4 *>(P.t_next + c4) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            vs[0] = fmaf(vs[0], an.x, tn.x);
            vs[1] = fmaf(vs[1], an.y, tn.y);
            vs[2] = fmaf(vs[2], an.z, tn.z);
            vs[3] = fmaf(vs[3], an.w, tn.w);
          }
          hf_half4 h4, l4;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            _Float16 hv, lv;
            hf_split_f16(vs[k], hv, lv, ovf_tile);
            h4[k] = hv;
            l4[k] = lv;
          }
          const long long unit = uofs[g] + (long long)(c4 >> 3) * oplane;
          *reinterpret_cast<hf_half4 *>(static_cast<char *>(P.oh) + unit * 16 + (c4 & 4) * 2) = h4;
          if (P.ol) *reinterpret_cast<hf_half4 *>(static_cast<char *>(P.ol) + unit * 16 + (c4 & 4) * 2) = l4;
        }
#endif
      }
    }
  }
#ifdef HF_WANT_F16_SPLIT
  if (split) hf_note_overflow(ovf_tile);
#endif
}

// the last block of a split-K tile, for the kernels whose blocks store through store_tile_rows (whole channel tiles)
template <int CT_TILES, int PG>
__device__ __forceinline__ void reduce_tile_rows(const ConvParams &P, const TileGeom &G, const GroupOfs &go, int co_wave,
                                                 int wave_pg, int li, int lh, int ty0, int tx0, int b0) {
  const int tw = 1 << G.lg_tw, th = 1 << G.lg_th;
  const long long oplane = (long long)P.out_h * P.out_w;
#pragma unroll
  for (int g = 0; g < PG; ++g) {
    const int p = (wave_pg + g) * 32 + li;
    const int px = p & (tw - 1), py = (p >> G.lg_tw) & (th - 1), im = p >> (G.lg_tw + G.lg_th);
    const int Y = ty0 + py, X = tx0 + px, b = b0 + im;
    if (!((Y < G.y0 + G.dh) && (X < G.x0 + G.dw) && (b < P.batch))) continue;
    const long long pix = (long long)Y * P.out_w + X;
#pragma unroll
    for (int ct = 0; ct < CT_TILES; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co_wave + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const long long i = go.o + ((long long)b * P.cout + co) * oplane + pix;
        P.out[i] = splitk_finish(P, i, b, go.c + co, pix, 0.0f, true);
      }
  }
}

template <int CT_TILES, int PG>
__device__ __forceinline__ void store_tile_rows(const ConvParams &P, const TileGeom &G, const GroupOfs &go,
                                                f32x16 (&acc)[1][CT_TILES][PG], int co_wave, int wave_pg, int li, int lh,
                                                int ty0, int tx0, int b0) {
  const bool partial = P.splits > 1 && !P.vsplit;
  const bool split = !partial && P.oh != nullptr, store32 = partial || P.out != nullptr;
  if (P.residual && !partial) {
    if (store32 && !split && P.act != ACT_QGELU)
      store_tile_rows_impl<CT_TILES, PG, true, 1>(P, G, go, acc, co_wave, wave_pg, li, lh, ty0, tx0, b0);
    else
      store_tile_rows_impl<CT_TILES, PG, true>(P, G, go, acc, co_wave, wave_pg, li, lh, ty0, tx0, b0);
    return;
  }
  if (!partial && P.act == ACT_QGELU) {
    if (split) store_tile_rows_impl<CT_TILES, PG, false>(P, G, go, acc, co_wave, wave_pg, li, lh, ty0, tx0, b0);
    else store_tile_rows_impl<CT_TILES, PG, false, 5>(P, G, go, acc, co_wave, wave_pg, li, lh, ty0, tx0, b0);
  } else if (store32 && !split) {
    store_tile_rows_impl<CT_TILES, PG, false, 1>(P, G, go, acc, co_wave, wave_pg, li, lh, ty0, tx0, b0);
  } else if (store32) {
    store_tile_rows_impl<CT_TILES, PG, false, 3>(P, G, go, acc, co_wave, wave_pg, li, lh, ty0, tx0, b0);
  } else if (split) {
    store_tile_rows_impl<CT_TILES, PG, false, 2>(P, G, go, acc, co_wave, wave_pg, li, lh, ty0, tx0, b0);
  }
}

inline int ilog2(int v) {
  int l = 0;
  while ((1 << (l + 1)) <= v) ++l;
  return l;
}
inline int pow2_floor(int v) { return 1 << ilog2(v); }
inline int pow2_ceil(int v) { return (v & (v - 1)) ? (pow2_floor(v) << 1) : v; }

// Tiles of `pt` pixels over a dh x dw OUTPUT domain (per image) of `batch` images.
// Prefers full 32-pixel rows; small planes put several images in one tile.
inline TileGeom make_geom(int y0, int x0, int dh, int dw, int batch, int pt, int first_block,
                          bool one_image = false, int rim = 2) {
  TileGeom g;
  g.y0 = y0; g.x0 = x0; g.dh = dh; g.dw = dw;
  int tw = pow2_ceil(dw);
  if (tw > 32 && dh > 1) tw = 32;  // rows of 32 consecutive pixels when the domain is 2-D
  if (tw > pt) tw = pt;
  int th = pow2_ceil(dh);
  if (th > pt / tw) th = pt / tw;
  if (one_image) {  // rim families of the pipelined kernel: stretch the tile instead of batching
    // images; two rows (columns) so that the halo tile stays within the staging budget
    if (dh == 1) { th = rim; tw = pt / rim; }
    else { tw = rim; th = pt / rim; }
  }
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
// LDS floats per staged channel; ext = extra rows/cols beyond (t-1)*stride + 1
inline int geom_xs(const TileGeom &g, int stride, int ext) {
  const int hp = ((1 << g.lg_th) - 1) * stride + 1 + ext, wp = ((1 << g.lg_tw) - 1) * stride + 1 + ext;
  return (1 << g.lg_nb) * hp * wp;
}


// convh.hip: 3x3 stride-1 same-resolution convolution (up: the transposed stride-2 one, written
// to the padded-pitch (2h+1)-row intermediate like hf_modconv3x3_up_f32) on v_mfma_f32_32x32x16_f16.
// nterms 3: fp32 operands split into fp16 (hi, lo) pairs, hi*hi + hi*lo + lo*hi in fp32
// accumulators (fp32-class accuracy, 5.3x the fp32 MFMA rate); nterms 1: plain fp16 operands.
// Returns HF_E_INVALID when the shape does not qualify.
int launch_conv_h(ConvParams &P, int nterms, bool up, const void *wt_hi, const void *wt_lo, hipStream_t st);
// convrow.hip: the row-pipeline form of a 32 -> 32 channel same-resolution layer with pre-split input (the 1024^2 conv);
// HF_E_INVALID when the layer does not qualify
int launch_conv_rows(ConvParams &P, int nterms, const void *wt_hi, const void *wt_lo, hipStream_t st);
extern thread_local int g_h_blocks;           // hf_debug_set_persistent_blocks: resident blocks the convh.hip grid is sized for (0 = 256 CUs)
extern thread_local int g_h_tune;             // hf_debug_set_tuning: bit 0 force early stage DMAs (convh.hip), bit 1 never the GEMM's 128-channel blocks (gemm_h.hip), ... (include/hairfast_hip.h)
extern int g_batch_invariant;                 // hf_set_batch_invariant (process-wide): plans (split-K counts, tile forms) from the per-sample shape only
// The batch count every decision that changes a sample's BITS is made with - the K partition (split-K factor) and the kernel
// family (fp32 split-K / tap-GEMM / tiled fp16-core kernels differ in summation order): the real one, or - in batch-invariant
// mode, where a sample's result must not depend on what it is batched with - a fixed canonical batch: 3, the Embedding stage's
// batch of a single swap, so that single swaps keep the plans they were tuned with.  TILE FORMS (64 x 128 / 256 / 512 pixels,
// 51 / 52, images per GEMM tile, pre-split or register-staged input) do not change the K order (bit-identical: tests/
// test_sim_encoders.py, test_sim_generator.py) and always follow the whole launch.
constexpr int kCanonBatch = 3;
inline int plan_batch(int batch) { return g_batch_invariant ? kCanonBatch : batch; }
extern thread_local int g_force_h;            // hf_debug_set_dispatch same_cfg 51/52: force the convh.hip tile configuration
void note_path(int path, int cfg);  // records what hf_debug_last_path reports
// split-K second pass (modconv.hip): out = epilogue(d * sum_z partial[z]), deterministic
int launch_splitk_reduce(ConvParams &P, bool with_epilogue, hipStream_t st);
// the registered arrival-counter buffer (hf_set_splitk_counters) when it can hold one counter per output tile
// (grid.x * grid.y) of this launch, else NULL (= the splitk_reduce launch)
unsigned int *splitk_counters_for(long long tiles);

}  // namespace hf_detail
