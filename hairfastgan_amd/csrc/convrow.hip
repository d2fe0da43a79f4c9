// convrow.hip - the generator's last 3x3 convolution (1024^2, 32 -> 32 channels, models/stylegan2/model.py:337-343 with the
// ToRGB of :356-362 fused) as a ROW PIPELINE on the fp16 matrix cores, f16x3 or plain fp16 operands (csrc/convh.hip).
//
// Why a kernel of its own: with K = 32 input channels a 512-pixel tile of the tiled kernel is two K stages - 6.9k MFMA
// cycles per SIMD against 136 KB of LDS-DMA per tile (at 100-185 issue cycles per KB), of which 36 KB are the SAME weights
// for every tile and a third of the rest is halo rows; the tile took 29k cycles (profiles/: 732 us per launch at batch 8,
// MFMA busy 0.32).  Here
//   * the weights live in REGISTERS: a wave's 36 A fragments (2 K chunks x 9 taps x hi / lo) are 144 VGPRs, loaded once
//     per block - no weight traffic and no weight fragment reads in the loop;
//   * a block owns a 64-column strip and walks DOWN it: input rows enter a ring of 18 row slots (66 pixels x 4 channel
//     blocks x hi / lo = 8.25 KB each) by LDS-DMA exactly once - no vertical halo re-reads, 3 % horizontal;
//   * one super-step = 8 output rows: wave w takes pixel tile (w & 1) of rows (w >> 1) and (w >> 1) + 4 as two interleaved
//     accumulation chains of 54 MFMAs; the eight rows of the NEXT super-step are requested during the first ten taps (one
//     copy per tap, between the MFMA groups) and waited for at the step's barrier (one barrier per 108 MFMAs);
//   * a step's epilogue runs after its barrier, at the head of the next step: no store is waited for fresh.
//   * ConvParams::rgb_skip (round 6): the epilogue finishes ToRGB - raw + bias + the x2-upsampled skip (hf_torgb_f32's
//     arithmetic, operation for operation) - instead of writing the raw product for a finishing launch (58 us and 200 MB
//     of traffic at batch 8): the skip rows a super-step needs (6 rows x 34 columns x 3 channels) travel through a 16-row
//     LDS ring like the input rows, two more 16-byte-per-lane copies per super-step.
// Accumulation order per output value = the tiled kernel's (chunk, tap, hh / hl / lh): identical bits.
// Measured at batch 8 (tools/probes/rows.py, bench.py): 732 -> 470 us per launch.  With the copies and the epilogue switched
// off (hf_debug_set_tuning bits 5 / 6) the MFMA loop alone takes 340 us against 184 us of pure MFMA time at 2.4 GHz; the
// copies add ~90 us and the epilogue ~60 us whether they are issued before the MFMA loop or between its MFMA groups, and
// whether the two waves of a SIMD issue them together or apart - the phases do not overlap further at one block per CU.
#include "conv_common.h"

using namespace hf_detail;

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

#ifndef HF_ROWS_PP
#define HF_ROWS_PP 0  // 1 = ping-pong super-steps (measured SLOWER: 507 -> 559 us per launch, generator 0->8 -2.7 %, profiles/r06bn_*: the layer is bound by its row stream, not by epilogues beside an idle pipe - the two phases halve the time a step's copies have to land)
#endif
#ifndef HF_ROWS_ILV
#define HF_ROWS_ILV HF_ROWS_PP  // (with ping-pong a computing wave is alone on its SIMD: its fragment reads go between its MFMAs)
#endif
constexpr int kSW = 64;                      // output columns of a strip
constexpr int kPXW = kSW + 2;                // staged columns (x0 - 1 .. x0 + 64)
constexpr int kPartUnits = 4 * kPXW;         // 16-byte units of one part (hi or lo) of a row slot: [channel block 4][66]
constexpr int kSlotUnits = 2 * kPartUnits;   // [part 2][channel block 4][66]
constexpr int kRing = 18;                    // row slots: 10 in use + 8 arriving
constexpr int kStep = 8;                     // output rows per super-step
constexpr int kDmaPerPart = (kPartUnits + 63) / 64;  // 5 (the fifth carries 8 lanes)
// finished ToRGB: skip rows in groups of four, [group 4][row 4][channel 3][10 units of 4 columns] floats - columns x0/2 - 4 ..
// x0/2 + 35 in whole 16-byte units (x0/2 is a multiple of 32, the plane width of 4: a unit lies inside the plane or outside) -
// group q = skip rows 4q+1 .. 4q+4 lives in slot (q + 1) & 3; then the 4x4 kernel and the three biases
constexpr int kSkCols = kSW / 2 + 8;          // 40 floats of a (row, channel): the 34 columns a strip reads sit at +3
constexpr int kSkRow = 3 * kSkCols;           // floats of one skip row (three channels)
constexpr int kSkGroup = 4 * kSkRow;          // 480 floats = 120 units: two copies of 64 lanes
constexpr int kSkFloats = 4 * kSkGroup + 16 + 4;

// NTERMS 3: f16x3 operands (hi, lo); 1: plain fp16 operands (BASELINE.json configs[4]) - no lo weights, no lo row parts, one
// MFMA per (tap, row) instead of three
template <int NTERMS>
__global__ __launch_bounds__(512, 2) void conv_rows_h(const ConvParams P, const _Float16 *__restrict__ wth,
                                                      const _Float16 *__restrict__ wtl, int rows_per_block, int segs, int ablate) {
  // ablate (hf_debug_set_tuning bits 5-7, timing experiments only): 1 no row copies, 2 no epilogue, 4 no MFMAs
  HF_DYN_LDS;
  half8 *ring = reinterpret_cast<half8 *>(hf_dyn_lds);                     // [kRing][kSlotUnits]
  float *ep = reinterpret_cast<float *>(ring + kRing * kSlotUnits);        // [32] d', [32] bias', [3][32] rgb weights
  float *skr = ep + 5 * 32;                                                 // P.rgb_skip: [4][kSkGroup] skip rows, [16] kernel, [3] bias
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int H = P.h, W = P.w;
  const long long plane = (long long)H * W;
  int bid = blockIdx.x;
  const int seg = bid % segs;
  bid /= segs;
  const int strips = W / kSW;
  const int strip = bid % strips, b = bid / strips;
  const int r0 = seg * rows_per_block, x0 = strip * kSW;
  const int nsteps = min(rows_per_block, H - r0) / kStep;

  // ---- weights: [chunk16][tap][kgroup 2][cout 32][8 halves] -> this lane's A fragments (co = li, kgroup = lh)
  constexpr int NPART = NTERMS == 3 ? 2 : 1;
  half8 ah[2][9], al[NTERMS == 3 ? 2 : 1][NTERMS == 3 ? 9 : 1];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const long long o = ((long long)((c * 9 + t) * 2 + lh) * 32 + li) * 8;
      ah[c][t] = *reinterpret_cast<const half8 *>(wth + o);
      if constexpr (NTERMS == 3) al[c][t] = *reinterpret_cast<const half8 *>(wtl + o);
    }
  if (tid < 32) {
    const float unscale = *reinterpret_cast<const float *>(wth + 9LL * 32 * 32);  // 2^-k of the weights' pre-scale (split_weights)
    ep[tid] = (P.d ? P.d[(long long)b * P.d_bstride + tid] : 1.0f) * unscale * P.scale;
    ep[32 + tid] = P.bias[tid] * P.scale;
    if (P.rgb_out) {
      const float sv = P.rgb_s[(long long)b * 32 + tid];
#pragma unroll
      for (int c = 0; c < 3; ++c) ep[64 + c * 32 + tid] = P.rgb_w[tid * 3 + c] * sv;
    }
    if (P.rgb_skip && tid < 19) skr[4 * kSkGroup + tid] = tid < 16 ? P.rgb_k4[tid] : (P.rgb_bias ? P.rgb_bias[tid - 16] : 0.0f);
  }

  // ---- LDS-DMA of one input row into a ring slot: per part 5 copies of 64 units; unit u = cb*66 + px of the part
  const char *xh_b = static_cast<const char *>(P.xh) + (long long)b * 4 * plane * 16;
  const char *xl_b = NTERMS == 3 ? static_cast<const char *>(P.xl) + (long long)b * 4 * plane * 16 : nullptr;
  int goff[kDmaPerPart];   // byte offset of the lane's unit inside the image at row 0; -1: no unit, -2: column outside the image
#pragma unroll
  for (int k = 0; k < kDmaPerPart; ++k) {
    const int u = 64 * k + lane;
    const int cb = u / kPXW, px = u - cb * kPXW, col = x0 - 1 + px;
    goff[k] = u >= kPartUnits ? -1 : ((col < 0 || col >= W) ? -2 : (int)(((long long)cb * plane + col) * 16));
  }
  const unsigned ring_addr = hf_lds_addr(ring);
  // copy `idx` (0 .. 9 = part * 5 + k) of input row y into ring slot `slot`
  auto dma_piece = [&](int idx, int y, int slot) {
    const int part = idx / kDmaPerPart, k = idx % kDmaPerPart;
    const bool inside = y >= 0 && y < H && goff[k] >= 0;
    unsigned off = inside ? (unsigned)goff[k] + (unsigned)y * (unsigned)W * 16u : 0u;
    const unsigned unit0 = (unsigned)(slot * kSlotUnits + part * kPartUnits + 64 * k);
    hf_glds16_raw_s_if(inside, part ? xl_b : xh_b, off, ring_addr + unit0 * 16u);
    if (!inside && goff[k] != -1) {  // zero padding: rows / columns outside the image
      half8 z;
#pragma unroll
      for (int q = 0; q < 8; ++q) z[q] = (_Float16)0.0f;
      ring[unit0 + lane] = z;
    }
  };
  auto dma_row = [&](int y, int slot) {
#pragma unroll
    for (int idx = 0; idx < NPART * kDmaPerPart; ++idx) dma_piece(idx, y, slot);
  };
  // finished ToRGB: waves 0 and 1 copy units 64w .. 64w+63 of a skip row group (16 bytes per lane, like the input rows)
  const int sh = H >> 1, sw = W >> 1;
  const float *skip_b = P.rgb_skip ? P.rgb_skip + (long long)b * 3 * sh * sw : nullptr;
  const unsigned skr_addr = hf_lds_addr(skr);
  auto dma_skip = [&](int q) {  // skip rows 4q+1 .. 4q+4 -> slot (q + 1) & 3
    if (wave >= 2) return;
    // the lane's (row of the group, channel, unit) recomputed per copy (opaque: no registers held across the MFMA loop)
    int u = 64 * wave + lane;
    HF_OPAQUE_I32(u);
    const int rq = u / 30, rem = u - rq * 30, ch = rem / 10, gcol = (x0 >> 1) - 4 + 4 * (rem - ch * 10);
    const int R = 4 * q + 1 + rq, slot = (q + 1) & 3;
    const bool unit = u < kSkGroup / 4, inside = unit && gcol >= 0 && gcol < sw && R >= 0 && R < sh;
    hf_glds16_raw_s_if(inside, skip_b, inside ? (unsigned)((ch * sh + R) * sw + gcol) * 4u : 0u,
                       skr_addr + (unsigned)(slot * kSkGroup + 256 * wave) * 4u);
    if (unit && !inside)  // rows / columns outside the plane
      *reinterpret_cast<float4 *>(skr + slot * kSkGroup + 4 * u) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  };
  const int sk_m = r0 / kStep;  // super-step S reads skip rows 4(m+S)-1 .. 4(m+S)+4: groups m+S-1 and m+S
  if (P.rgb_skip) {
    dma_skip(sk_m - 1);
    dma_skip(sk_m);
  }
  // prologue: input rows r0-1 .. r0+8 -> slots 0 .. 9
  dma_row(r0 - 1 + wave, wave);
  if (wave < 2) dma_row(r0 + 7 + wave, 8 + wave);
  hf_barrier_keep_young<0>();

  const int j = wave & 1, rw = wave >> 1;   // pixel tile of the strip, row of the half super-step
  const int X = x0 + 32 * j + li;
  const float nws = P.noise ? P.noise_w[0] * P.scale : 0.0f;
  f32x16 acc[2];

  // One super-step of this wave: output rows `ro` and `ro + 4` (relative to r0) of its pixel tile, as TWO interleaved
  // accumulation chains - consecutive MFMAs never depend on each other (a single chain ran the matrix pipe at 54 %: a
  // dependent MFMA issues only after its predecessor's eight passes have drained) - each in the tiled kernel's order.
  // Input row ro + ky - 1 sits in slot (ro + ky) % 18.  `dma`: the wave's ten copies of the next super-step's row
  // (y_next -> slot_next) are issued one per tap, between the MFMA groups.
  auto compute = [&](int ro, int base18, bool dma, int y_next, int slot_next) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[r][q] = 0.0f;
    int srow[2][3];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        int s = base18 + (ro & 7) + 4 * r + ky;   // base18 = (8 * step) % 18, ro & 7 = row inside the super-step
        s = s >= kRing ? s - kRing : s;
        srow[r][ky] = s * kSlotUnits + 32 * j + li;
      }
    // fragments of tap i+1 are fetched before the MFMAs of tap i (two register slots): the LDS latency runs under them
    half8 bh[2][2], bl[NTERMS == 3 ? 2 : 1][2];  // [slot][row]
    auto fetch = [&](int i, int slot) {
      const int c = i / 9, t = i % 9;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int u = srow[r][t / 3] + (2 * c + lh) * kPXW + t % 3;
        bh[slot][r] = ring[u];
        if constexpr (NTERMS == 3) bl[slot][r] = ring[u + kPartUnits];
      }
    };
    fetch(0, 0);
#pragma unroll
    for (int i = 0; i < 18; ++i) {
      const int c = i / 9, t = i % 9, sl = i & 1;
      if (i + 1 < 18) fetch(i + 1, sl ^ 1);
      if (ablate & 4) continue;
      if (dma && !(ablate & 1) && i < NPART * kDmaPerPart) dma_piece(i, y_next, slot_next);
      if (!HF_ROWS_ILV) __builtin_amdgcn_sched_barrier(0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[c][t], bh[sl][0], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[c][t], bh[sl][1], acc[1], 0, 0, 0);
      if constexpr (NTERMS == 3) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[c][t], bl[sl][0], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[c][t], bl[sl][1], acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[c][t], bh[sl][0], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[c][t], bh[sl][1], acc[1], 0, 0, 0);
      }
      if constexpr (HF_ROWS_ILV != 0) {  // the next tap's 2 (4) fragment reads between this tap's MFMAs (A/B: see HF_H_ILV, csrc/convh.hip)
        if (i + 1 < 18) {
#pragma unroll
          for (int k = 0; k < 2 * NPART; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, (NTERMS == 3 ? 6 : 2) - 2 * NPART, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // epilogue of the pass held in acc: v = lrelu(acc * d + noise_w * noise + bias) * scale (positively homogeneous: the
  // scale is folded into d, bias and the noise weight, as in the tiled kernel's fast epilogue), fp32 output and / or the
  // fused ToRGB partial sums.  D layout: channel = (q & 3) + 8 * (q >> 2) + 4 * lh, pixel = li.
  auto epilogue = [&](const f32x16 &av, int ro, float noise_v) {
    if (ablate & 2) return;
    int lh_o = lh;
    HF_OPAQUE_I32(lh_o);
    const int Y = r0 + ro;
    float nzv = nws * noise_v;
    HF_OPAQUE_F32(nzv);  // a product of its own (no contraction into the add): bit-equal to the tiled kernel
    const long long pix = (long long)Y * W + X;
    float rgb[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c4 = 8 * q + 4 * lh_o;
      const float4 dm = *reinterpret_cast<const float4 *>(ep + c4), bs = *reinterpret_cast<const float4 *>(ep + 32 + c4);
      const float dmv[4] = {dm.x, dm.y, dm.z, dm.w}, bsv[4] = {bs.x, bs.y, bs.z, bs.w};
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float o = fmaf(av[4 * q + k], dmv[k], nzv + bsv[k]);
        v[k] = fmaxf(o, o * P.alpha);
      }
      if (P.out) {
        float *ob = P.out + ((long long)b * 32 + c4) * plane + pix;
#pragma unroll
        for (int k = 0; k < 4; ++k) ob[(long long)k * plane] = v[k];
      }
      if (P.rgb_out) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float4 wv = *reinterpret_cast<const float4 *>(ep + 64 + c * 32 + c4);
          rgb[c] = fmaf(v[3], wv.w, fmaf(v[2], wv.z, fmaf(v[1], wv.y, fmaf(v[0], wv.x, rgb[c]))));
        }
      }
    }
    if (P.rgb_out) {
#pragma unroll
      for (int c = 0; c < 3; ++c) rgb[c] += __shfl_xor(rgb[c], 32, 64);
      if (P.rgb_skip) {
        // hf_torgb_f32 with one slab: r = raw + bias; up = the 2 x 2 taps of the zero-insert x2 upsampler (pad (2, 1), 4x4 true
        // convolution) that hit samples, rows iy0 / iy0 + 1 and columns ix0 / ix0 + 1 in that order; out = r + up.  Rows and
        // columns outside the plane are zeros in the ring (fmaf(0, k, up) = up: the reference skips them).
        int Yo = Y, li_o = li;  // (opaque: nothing of this block is hoisted out of the step loop into registers that stay live)
        HF_OPAQUE_I32(Yo);
        HF_OPAQUE_I32(li_o);
        const int ky0 = Yo & 1, kx0 = li_o & 1;
        const int iy0 = (Yo + ky0 - 2) >> 1;
        const int col0 = 16 * j + (li_o >> 1) + kx0 + 3;  // column ix0 - (x0/2 - 4)
        const float *s0 = skr + ((iy0 + 3) & 15) * kSkRow + col0, *s1 = skr + ((iy0 + 4) & 15) * kSkRow + col0;
        const float *k4 = skr + 4 * kSkGroup;
        const float k00 = k4[(3 - ky0) * 4 + 3 - kx0], k01 = k4[(3 - ky0) * 4 + 1 - kx0];
        const float k10 = k4[(1 - ky0) * 4 + 3 - kx0], k11 = k4[(1 - ky0) * 4 + 1 - kx0];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float r = rgb[c] + k4[16 + c];
          float up = fmaf(s0[c * kSkCols], k00, 0.0f);
          up = fmaf(s0[c * kSkCols + 1], k01, up);
          up = fmaf(s1[c * kSkCols], k10, up);
          up = fmaf(s1[c * kSkCols + 1], k11, up);
          rgb[c] = r + up;
        }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c)
        if (lh_o == 0) P.rgb_out[((long long)b * 3 + c) * plane + pix] = rgb[c];
    }
  };

  int base18 = 0;
  float nzp[2] = {0.0f, 0.0f};
  for (int S = 0; S < nsteps; ++S) {
    // the next super-step's eight new input rows r0 + 8S + 9 .. + 16 -> slots (8S + 10 + w) % 18, one row per wave
    const bool more = S + 1 < nsteps;
    int slot_next = base18 + 10 + wave;
    slot_next = slot_next >= kRing ? slot_next - kRing : slot_next;
    const int y_next = r0 + kStep * S + 9 + wave;
    const int ro = kStep * S + rw;
    float nz0 = 0.0f, nz1 = 0.0f;
    if (P.noise) {
      const float *np = P.noise + (long long)b * P.noise_bstride + (long long)(r0 + ro) * W + X;
      nz0 = np[0];
      nz1 = np[4LL * W];
    }
    if constexpr (HF_ROWS_PP != 0) {
      // Ping-pong (round 6, as the tiled kernels' K loops: csrc/convh.hip lesson 16): the two waves of a SIMD take TURNS on the
      // matrix pipe.  Phase A: waves 0-3 run their 108 MFMAs alone on their SIMDs while waves 4-7 run the PREVIOUS step's
      // epilogue and issue their row copies; role-swap barrier; phase B: waves 4-7 compute, waves 0-3 run THIS step's epilogue
      // (their accumulators are complete) and issue their copies; end-of-step barrier (copies landed).  Every wave's epilogue
      // falls into its idle phase - in the one-phase form all eight waves ran theirs side by side behind the barrier with the
      // pipe idle (a step took ~2.3x its MFMA time).  Same arithmetic per wave: equal bits.
      if (wave < 4) {
        compute(ro, base18, false, y_next, slot_next);
        hf_barrier_lds();
        epilogue(acc[0], ro, nz0);
        epilogue(acc[1], ro + 4, nz1);
        if (more && !(ablate & 1)) dma_row(y_next, slot_next);
        if (more && P.rgb_skip) dma_skip(sk_m + S + 1);
      } else {
        if (S > 0) {
          epilogue(acc[0], ro - kStep, nzp[0]);
          epilogue(acc[1], ro - kStep + 4, nzp[1]);
        }
        if (more && !(ablate & 1)) dma_row(y_next, slot_next);
        hf_barrier_lds();
        compute(ro, base18, false, y_next, slot_next);
        nzp[0] = nz0;
        nzp[1] = nz1;
      }
      hf_barrier_keep_young<0>();  // the new rows have landed, the oldest eight slots are free
      base18 += kStep;
      base18 = base18 >= kRing ? base18 - kRing : base18;
      continue;
    }
    // the two waves of a SIMD (w, w + 4) issue their copies at different times: one before its epilogue, the other between
    // the MFMA groups of its first ten taps - each stalls on the copy issue while its partner has MFMAs to issue
    if (more && wave < 4 && !(ablate & 1)) dma_row(y_next, slot_next);
    if (more && P.rgb_skip) dma_skip(sk_m + S + 1);  // the next super-step's new skip rows (waited for at this step's barrier)
    if (S > 0) {  // the previous super-step's epilogue: its stores drain under this step's MFMAs, none is waited for fresh
      epilogue(acc[0], ro - kStep, nzp[0]);
      epilogue(acc[1], ro - kStep + 4, nzp[1]);
    }
    compute(ro, base18, more && wave >= 4, y_next, slot_next);
    nzp[0] = nz0;
    nzp[1] = nz1;
    hf_barrier_keep_young<0>();  // the new rows have landed, the oldest eight slots are free
    base18 += kStep;
    base18 = base18 >= kRing ? base18 - kRing : base18;
  }
  if (nsteps > 0 && !(HF_ROWS_PP != 0 && wave < 4)) {  // (ping-pong: waves 0-3 ran their last epilogue inside the loop)
    epilogue(acc[0], kStep * (nsteps - 1) + rw, nzp[0]);
    epilogue(acc[1], kStep * (nsteps - 1) + rw + 4, nzp[1]);
  }
}

}  // namespace

namespace hf_detail {

// The row-pipeline form of a same-resolution 3x3 layer (launch_conv_h's contract): HF_E_INVALID when the layer does not qualify.
int launch_conv_rows(ConvParams &P, int nterms, const void *wth, const void *wtl, hipStream_t st) {
  if (P.cin != 32 || P.cout != 32 || !P.xh || (nterms == 3 && !P.xl) || P.s || P.t || P.oh || P.residual || P.groups > 1 ||
      P.stride != 1 || (P.w % kSW) || (P.h % kStep) || P.out_h != P.h || P.out_w != P.w || !wth || (nterms == 3 && !wtl) ||
      (nterms != 1 && nterms != 3))
    return HF_E_INVALID;
  if (!P.bias || P.act != ACT_LRELU || !(P.alpha >= 0.0f && P.alpha <= 1.0f) || !(P.scale > 0.0f)) return HF_E_INVALID;
  if ((!P.out && !P.rgb_out) || (P.rgb_out && (!P.rgb_w || !P.rgb_s || P.rgb_slabs != 1))) return HF_E_INVALID;
  if (P.rgb_skip && (!P.rgb_out || !P.rgb_k4 || (P.h & 1) || (P.w & 1) || ((size_t)P.rgb_skip & 15))) return HF_E_INVALID;  // (16-byte skip units)
  if ((long long)4 * P.h * P.w * 16 >= (1LL << 31)) return HF_E_INVALID;  // 32-bit byte offsets inside an image
  const int strips = P.w / kSW;
  // vertical segments: LDS allows one block per CU, so the launch runs in ceil(blocks / 256) rounds of rows_per_block rows
  // (+ ~24 rows' worth of prologue: weights, the first ten input rows): the power-of-two split with the cheapest schedule.
  // Batch 8 at 1024^2: 2 segments = 256 blocks, one round (measured 2-3 % faster than 512 blocks in two rounds).
  int segs = 1;
  long long best = -1;
  for (int s2 = 1; (P.h / s2) % kStep == 0 && P.h / s2 >= 4 * kStep; s2 *= 2) {
    const long long rounds = ((long long)P.batch * strips * s2 + 255) / 256;
    const long long cost = rounds * (P.h / s2 + 24);
    if (best < 0 || cost < best) {
      best = cost;
      segs = s2;
    }
    if (P.h % (s2 * 2)) break;
  }
  const int rows_per_block = P.h / segs;
  const long long blocks = (long long)P.batch * strips * segs;
  if (blocks >= (1LL << 31)) return HF_E_INVALID;
  const size_t lds = (size_t)kRing * kSlotUnits * 16 + 5 * 32 * sizeof(float) + (P.rgb_skip ? kSkFloats * sizeof(float) : 0);
  if (nterms == 3)
    hipLaunchKernelGGL(conv_rows_h<3>, dim3((unsigned)blocks), dim3(512), lds, st, P, static_cast<const _Float16 *>(wth),
                       static_cast<const _Float16 *>(wtl), rows_per_block, segs, (g_h_tune >> 5) & 7);
  else
    hipLaunchKernelGGL(conv_rows_h<1>, dim3((unsigned)blocks), dim3(512), lds, st, P, static_cast<const _Float16 *>(wth),
                       static_cast<const _Float16 *>(wtl), rows_per_block, segs, (g_h_tune >> 5) & 7);
  return hf_launch_status();
}

}  // namespace hf_detail
