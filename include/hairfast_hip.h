/*
 * hairfast_hip.h - C ABI of libhairfast_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the HairFastGAN generator hot path.  Every entry point
 * takes raw DEVICE pointers (fp32, NCHW contiguous unless stated), plain ints
 * and a hipStream_t passed as void*; none allocates, synchronises or touches
 * torch types.  Return value: 0 on success, a negative HF_E_* code otherwise
 * (hf_strerror() gives the text).  Kernels are enqueued asynchronously on the
 * given stream; outputs never alias inputs.
 *
 * Each entry point cites the reference interface it replaces
 * (paths relative to the HairFastGAN tree).
 */
#ifndef HAIRFAST_HIP_H
#define HAIRFAST_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define HF_OK 0
#define HF_E_INVALID (-1)   /* bad argument (null pointer, non-positive dim, unsupported size) */
#define HF_E_LAUNCH (-2)    /* hipGetLastError() reported a launch failure */
#define HF_E_WORKSPACE (-3) /* caller-provided workspace too small */

/* OR-ed into the `act` argument of hf_conv2d_f32 / hf_conv2d_f16_f32: add `residual` BEFORE the activation
 * (ResNet BasicBlock: relu(shortcut + bn(conv)), resnet.py:41-45) instead of after it (IBasicBlock) */
#define HF_ACT_RESIDUAL_FIRST 16

const char *hf_strerror(int code);
/* ABI version of this header; bumped on any signature change. */
int hf_abi_version(void);

/* ---------------------------------------------------------------------------
 * upfirdn2d: upsample (zero insert) -> pad/crop -> FIR (true convolution) ->
 * downsample, on `major` independent [in_h, in_w] planes (major = N*C).
 * Replaces: models/stylegan2/op/upfirdn2d.cpp:12-19 `upfirdn2d(input, kernel,
 * up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1)` with input
 * viewed as [major, in_h, in_w, 1] (op/upfirdn2d.py:99), and the kernels in
 * models/stylegan2/op/upfirdn2d_kernel.cu:49-207.
 * out must hold major*out_h*out_w floats, out_h = (in_h*up_y+pad_y0+pad_y1-kh)/down_y+1.
 * kernel: device pointer, [kh, kw] row-major, kh,kw <= 8.
 */
int hf_upfirdn2d_f32(float *out, const float *in, const float *kernel, int major, int in_h, int in_w,
                     int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1,
                     int pad_y0, int pad_y1, void *stream);

/* ---------------------------------------------------------------------------
 * y = leaky_relu(x + bias[(i / step_b) % n_bias], alpha) * scale
 * Replaces: models/stylegan2/op/fused_bias_act.cpp:11-17 `fused_bias_act(input,
 * bias, refer, act=3, grad=0, alpha, scale)` (fused_bias_act_kernel.cu:18-49,
 * case 30; step_b as :69-71).  n elements; bias may be NULL (n_bias ignored).
 */
int hf_fused_bias_act_f32(float *out, const float *x, const float *bias, long long n, int n_bias,
                          int step_b, float alpha, float scale, void *stream);

/* ---------------------------------------------------------------------------
 * Noise injection + FusedLeakyReLU in one pass:
 *   y[b,c,p] = leaky_relu(x[b,c,p] + noise_w[0]*noise[b*noise_bstride + p] + bias[c], alpha) * scale
 * Replaces: NoiseInjection.forward + FusedLeakyReLU.forward
 * (models/stylegan2/model.py:288-293, op/fused_act.py:73-96).  noise_w is a
 * DEVICE pointer to the scalar parameter; noise_bstride = 0 broadcasts one
 * [1,1,H,W] map over the batch, H*W for per-sample noise.
 */
int hf_noise_bias_act_f32(float *out, const float *x, const float *noise, const float *noise_w,
                          const float *bias, int batch, int channels, int hw, long long noise_bstride,
                          float alpha, float scale, void *stream);

/* ---------------------------------------------------------------------------
 * One-time re-layout of a ModulatedConv2d weight (frozen parameters):
 *   wt[tap][ci][co]  = scale * weight[0][co][ci][tap],  scale = 1/sqrt(cin*k*k)
 *   wsq[co][ci]      = sum_tap (scale * weight[0][co][ci][tap])^2
 * weight: reference layout [1, cout, cin, k, k] (models/stylegan2/model.py:223-225,
 * scale :219-220).  k = 1 or 3.  wsq may be NULL (ToRGB has no demodulation).
 */
int hf_modconv_prepare_f32(float *wt, float *wsq, const float *weight, int cout, int cin, int k,
                           void *stream);

/* ---------------------------------------------------------------------------
 * Modulation: s[b,ci] = sum_j latent[b*lat_stride + j] * (mod_w[ci,j] / sqrt(style_dim)) + mod_b[ci]
 * Replaces: EqualLinear.forward of ModulatedConv2d.modulation
 * (models/stylegan2/model.py:153-163 with lr_mul=1, called at :241).
 * `latent` may be a strided row view of W+ (latent[:, i]; lat_stride = n_latent*style_dim).
 */
int hf_modulation_f32(float *s, const float *latent, long long lat_stride, const float *mod_w,
                      const float *mod_b, int batch, int cin, int style_dim, void *stream);

/* Demodulation coefficients: d[b,co] = rsqrt(sum_ci wsq[co,ci] * s[b,ci]^2 + 1e-8)
 * Replaces: models/stylegan2/model.py:244-246 (algebraically equal: the
 * per-sample weight is scale*W*s, so its squared norm factors through wsq). */
int hf_demod_f32(float *d, const float *s, const float *wsq, int batch, int cin, int cout, void *stream);

/* Range normalisation of one layer's (s, d) pair, IN PLACE: per sample b, with
 * e = floor(log2 max_ci |s[b,ci]|):  s[b,:] *= 2^-e,  d[b,:] *= 2^e   (max|s| ends up in [1, 2)).
 * The demodulated convolution y = d * sum w (s x) is invariant under it and a power of two is exact
 * in fp32, so every fp32 consumer returns bit-identical results; the fp16 (hi, lo) operand split of
 * s*x in the matrix-core modes (hf_modconv3x3_f16_* below) then cannot overflow through a large
 * trained style and does not lose its lo part to fp16 subnormals through a small one.  No reference
 * counterpart (the reference is fp32 throughout); call it after hf_demod_f32.  hf_style_batch_f32
 * applies it to every job that has a demodulation. */
int hf_style_normalize_f32(float *s, float *d, int batch, int cin, int cout, void *stream);

/* ---------------------------------------------------------------------------
 * Fused StyledConv, same resolution (3x3, pad 1):
 *   y[b,co] = act( d[b,co] * sum_{ci,tap} wt[tap,ci,co] * (s[b,ci]*x[b,ci] shifted by tap)
 *                  + noise_w*noise + bias[co] ) * sqrt2-scale
 * Replaces: ModulatedConv2d.forward (models/stylegan2/model.py:238-250,273-277)
 * + NoiseInjection + FusedLeakyReLU (StyledConv.forward :337-343).  The
 * modulation is applied to the activations and the demodulation to the
 * outputs instead of materialising per-sample weights; fp32 MFMA accumulate.
 * s, d: from hf_modulation_f32 / hf_demod_f32 (d NULL = no demodulation;
 * s NULL = plain convolution).  noise NULL = no noise term; bias NULL = no
 * bias/activation epilogue (raw conv output, alpha/scale ignored).
 * workspace: see hf_modconv_workspace_floats.
 */
int hf_modconv3x3_f32(float *out, const float *x, const float *wt, const float *s, const float *d,
                      const float *noise, const float *noise_w, long long noise_bstride,
                      const float *bias, int batch, int cin, int cout, int h, int w, float alpha,
                      float scale, float *workspace, long long workspace_floats, void *stream);

/* All style GEMVs of one generator forward in two launches: for every job (= one ModulatedConv2d)
 *   s[b, ci] = hf_modulation_f32(latent[b, style_row, :], mod_w, mod_b)     -> out[s_ofs + b*cin + ci]
 *   d[b, co] = hf_demod_f32(s, wsq)   (jobs with wsq != NULL)                -> out[d_ofs + b*cout + co]
 * (EqualLinear of the 26 modulations + the 17 demodulations, models/stylegan2/model.py:241-246;
 * SURVEY section 8 row a10).  jobs: device array; latent: [batch, n_latent, style_dim] with the
 * given strides (floats).  Same arithmetic per element as the single-layer entry points. */
typedef struct hf_style_job {
  const float *mod_w;   /* [cin, style_dim] */
  const float *mod_b;   /* [cin] */
  const float *wsq;     /* [cout, cin] from hf_modconv_prepare_f32, NULL = no demodulation */
  int cin, cout;
  int style_row, reserved;
  long long s_ofs, d_ofs; /* float offsets into out */
} hf_style_job;
int hf_style_batch_f32(float *out, const float *latent, long long lat_bstride, long long lat_rstride,
                       const hf_style_job *jobs, int n_jobs, int batch, int style_dim, int max_cin, int max_cout,
                       void *stream);

/* ---------------------------------------------------------------------------
 * fp16-matrix-core variant of hf_modconv3x3_f32 (same contraction, same epilogue, fp32
 * tensors in HBM, fp32 accumulation; v_mfma_f32_32x32x16_f16).  Reference counterpart:
 * the fp16 autocast path the headline numbers of BASELINE.json configs[4] name; the
 * reference itself runs ModulatedConv2d in fp32 (models/stylegan2/model.py:238-277).
 *   nterms 3 ("f16x3"): operands split into fp16 (hi, lo) pairs, a*b = hi*hi + hi*lo + lo*hi,
 *            each MFMA product exact in the fp32 accumulator, at 3/16 of the fp32 MFMA time.
 *            Accuracy: an operand v is carried with relative error <= 2^-22 while |v| >= 2^-3 and
 *            with ABSOLUTE error <= 2^-25 below that (lo becomes an fp16 subnormal); the dropped
 *            lo*lo term is <= 2^-22 relative.  Weights are pre-scaled by a power of two so that
 *            max|w| sits at 2^13 (every weight within 2^-15 of the largest keeps 22 bits), styles
 *            are normalised to max|s| in [1,2) (hf_style_normalize_f32), so a dot product's large
 *            terms are exact to ~2e-7 and its small terms to 3e-8 * |w| absolute: the same class
 *            as an fp32 reassociation for activations of magnitude >= ~1e-2.
 *            Range: full accuracy for |s*x| < 65504 (i.e. |x| < 32752 after the style normalisation);
 *            both parts saturate at 65504 instead of becoming inf/NaN, so values up to 131008 are
 *            still carried (to 2^-11 of their excess); larger ones clamp and are counted by
 *            hf_f16_overflow_count().
 *   nterms 1 ("f16"): operands rounded to fp16 (rel. error ~5e-4 per product), saturating at 65504.
 * wt_hi / wt_lo: from hf_conv_split_weights_f16, layout [cin/16][tap][2][cout][8] fp16.
 * wt_hi must hold 9*cin*cout fp16 values PLUS A 16-BYTE TRAILER (float 2^-k of the weights'
 * power-of-two pre-scale, read by the conv kernels and folded into the output scale; 12 bytes
 * reserved); wt_lo 9*cin*cout values, may be NULL for nterms 1.
 * Shapes: cin % 16 == 0, w >= 32 and (cout % 64 == 0, h >= 8) or (cout % 32 == 0, h >= 16);
 * anything else returns HF_E_INVALID and the caller uses hf_modconv3x3_f32.
 */
int hf_conv_split_weights_f16(void *wt_hi, void *wt_lo, const float *wt, int cin, int cout, void *stream);
/* The same for `taps` in {9, 1}: prepared weights [taps][cin][cout] -> [cin/16][tap][2][cout][8] fp16 hi / lo (+ the
 * 16-byte trailer behind hi).  taps = 1: the weights of hf_conv1x1_f16_f32 (from hf_conv_prepare_f32 with k = 1). */
int hf_conv_split_weights_f16_taps(void *wt_hi, void *wt_lo, const float *wt, int cin, int cout, int taps, void *stream);
/* Elements the fp16 split had to clamp (|v| > 131008 or NaN) since the last reset, summed over
 * all kernels of the library.  SYNCHRONOUS (copies a device counter; do not call inside a stream
 * capture).  A non-zero value means a tensor left the fp16-pair range: re-run with the exact
 * fp32 kernels (hf_modconv3x3_f32 / hf_modconv3x3_up_f32).  reset != 0 clears the counter. */
long long hf_f16_overflow_count(int reset);
int hf_modconv3x3_f16_f32(float *out, const float *x, const void *wt_hi, const void *wt_lo, int nterms,
                          const float *s, const float *d, const float *noise, const float *noise_w,
                          long long noise_bstride, const float *bias, int batch, int cin, int cout, int h,
                          int w, float alpha, float scale, void *stream);
/* hf_modconv3x3_f16_f32 with ToRGB's 1x1 modulated conv fused into the epilogue (cout % 32 == 0).  A wave holds
 * 64 (cout % 64 == 0) or 32 output channels of its pixels and writes their part of the sum into its own slab:
 *   rgb_raw[b, j*3+c, Y, X] = sum_{co in slab j} rgb_wt[co*3+c] * rgb_s[b*cout+co] * out[b,co,Y,X],
 *   rgb_raw: [batch][3*slabs][h][w], slabs = hf_modconv3x3_f16_rgb_slabs(cout)  (1 for cout 32 / 64)
 * i.e. ToRGB.forward (models/stylegan2/model.py:356-362) before its bias and upsampled skip; finish with
 * hf_torgb_f32(out_rgb, rgb_raw, I, NULL, bias, skip, k4, batch, 3*slabs, h, w), I = `slabs` stacked 3x3 identities
 * (a fixed summation order: deterministic).  rgb_wt: ToRGB's prepared 1x1 weight ([cout][3], from
 * hf_modconv_prepare_f32), rgb_s: its modulation (hf_modulation_f32).  Saves re-reading the layer's output
 * (4*cout*H*W bytes) - and writing it, when nothing else consumes it (hf_modconv3x3_f16_pre_f32 with out = NULL).
 * More than one slab needs the standard tail (bias != NULL); otherwise HF_E_INVALID. */
int hf_modconv3x3_f16_rgb_slabs(int cout);
int hf_modconv3x3_f16_rgb_f32(float *out, const float *x, const void *wt_hi, const void *wt_lo, int nterms,
                              const float *s, const float *d, const float *noise, const float *noise_w,
                              long long noise_bstride, const float *bias, int batch, int cin, int cout, int h,
                              int w, float alpha, float scale, float *rgb_raw, const float *rgb_wt,
                              const float *rgb_s, void *stream);
/* hf_modconv3x3_f16_f32 whose input arrives pre-split from its producer
 * (hf_blur_noise_bias_act_split_f16: x_hi / x_lo = fp16 (hi, lo) of s*x, [batch][cin/8][h][w][8]):
 * the kernel stages weights AND activations by LDS-DMA - no per-element loads, no conversion, and
 * no s argument (already applied).  rgb_raw / rgb_wt / rgb_s as in hf_modconv3x3_f16_rgb_f32, all
 * NULL = no fused ToRGB.  x_lo may be NULL for nterms 1.  out may be NULL when rgb_raw is given (the
 * generator's last layer: nothing but its ToRGB reads the activation - 4*cout*H*W bytes not written).
 * Same shapes as hf_modconv3x3_f16_f32. */
int hf_modconv3x3_f16_pre_f32(float *out, const void *x_hi, const void *x_lo, const void *wt_hi, const void *wt_lo,
                              int nterms, const float *d, const float *noise, const float *noise_w,
                              long long noise_bstride, const float *bias, int batch, int cin, int cout, int h, int w,
                              float alpha, float scale, float *rgb_raw, const float *rgb_wt, const float *rgb_s,
                              void *split_hi, void *split_lo, const float *s_next, void *stream);
/* ABI 13: the generator's LAST StyledConv (models/stylegan2/model.py:337-343) with its ToRGB (:356-365) complete - the 1024^2
 * layer, cin = cout = 32, on pre-split input: image[b, c, Y, X] = sum_co rgb_wt[co*3+c] * rgb_s[b*cout+co] * out[b,co,Y,X]
 * + rgb_bias[c] + Upsample(skip)[b, c, Y, X]  (skip [batch][3][h/2][w/2], kernel4x4 = make_kernel([1,3,3,1]) * 4: the
 * upfirdn2d up = 2, pad (2, 1) of :49-53), operation for operation what hf_modconv3x3_f16_pre_f32 (rgb_raw, one slab) followed
 * by hf_torgb_f32 on that slab computes - without the raw product in memory and without the second launch.  Row-pipeline
 * shapes only (w a multiple of 64, h of 8; not under hf_debug_set_dispatch / tuning bit 4): HF_E_INVALID otherwise - callers
 * fall back to the two calls above. */
int hf_modconv3x3_f16_pre_image_f32(float *image, const void *x_hi, const void *x_lo, const void *wt_hi, const void *wt_lo,
                                    int nterms, const float *d, const float *noise, const float *noise_w,
                                    long long noise_bstride, const float *bias, int batch, int cin, int cout, int h, int w,
                                    float alpha, float scale, const float *rgb_wt, const float *rgb_s, const float *rgb_bias,
                                    const float *skip, const float *kernel4x4, void *stream);
/* split_hi / split_lo (NULL = off): the epilogue ALSO writes s_next[b,co] * out as fp16 (hi, lo) pairs,
 * K-blocked [batch][cout/8][h][w][8] - the pre-split input of the next layer's transposed conv
 * (hf_modconv3x3_up_f16_pre_f32); out may then be NULL too if nobody reads the fp32 activation.
 *
 * hf_modconv3x3_up_f16_f32 on pre-split input (x_hi / x_lo of s*x from the call above). */
int hf_modconv3x3_up_f16_pre_f32(float *tmp, const void *x_hi, const void *x_lo, const void *wt_hi, const void *wt_lo,
                                 int nterms, const float *d, int batch, int cin, int cout, int h, int w, int tmp_pitch,
                                 void *stream);
/* Part 1 of the upsampling StyledConv (hf_modconv3x3_up_f32) on the fp16 matrix cores: same
 * intermediate [batch, cout, 2h+1, tmp_pitch], same weights as hf_modconv3x3_f16_f32 (the
 * transposed conv's tap flip is in the phase mapping, not in the layout).  Shapes:
 * cin % 16 == 0, cout % 32 == 0, h*w >= 256 (cout % 64 == 0) or >= 512; otherwise HF_E_INVALID. */
int hf_modconv3x3_up_f16_f32(float *tmp, const float *x, const void *wt_hi, const void *wt_lo, int nterms,
                             const float *s, const float *d, int batch, int cin, int cout, int h, int w,
                             int tmp_pitch, void *stream);

/* The whole upsampling StyledConv in ONE kernel (csrc/convh.hip, FUSE): transposed 3x3 conv (stride 2) of
 * s*x on the fp16 matrix cores (f16x3 operands), demodulation, 4x4 blur with pad (1,1), noise, bias, leaky
 * ReLU - models/stylegan2/model.py:252-263 + :337-343 - without the [batch,cout,2h+1,2w+1] intermediate
 * that hf_modconv3x3_up_f16_f32 writes and hf_blur_noise_bias_act_*_f32 reads back (8 bytes of HBM traffic
 * per intermediate element, ~1 ms of a batch-8 1024^2 forward).  Blocks compute overlapping tiles of
 * 16 x 32 phase-domain positions and emit the blurred output of the 14 x 30 interior ones.
 *   x (fp32, with s) or x_hi/x_lo (pre-split, s already applied; x, s unused)      input [batch,cin,h,w]
 *   out (fp32 [batch,cout,2h,2w]) and / or split_hi/split_lo (fp16 pairs of s_next*out, K-blocked
 *   [batch][cout/8][2h][2w][8], for a pre-split consumer; s_next NULL = 1) - at least one of the two.
 *   blur_k1d_x / blur_k1d_y: HOST pointers to the 4 taps of the two 1-D factors of the blur kernel
 *   (kernel4x4[r][j] = blur_k1d_y[r] * blur_k1d_x[j]; the module's `blur.kernel` is such an outer product,
 *   model.py:24-32): the fused form needs a separable kernel - callers with a general 4x4 kernel use the
 *   two-pass entry points.  cin % 16 == 0, cout % 32 == 0.  Operand mode: wt_lo != NULL = f16x3 (three MFMAs per product);
 *   wt_lo NULL = plain fp16 operands (BASELINE.json configs[4]) - x_lo and split_lo are then not read / written. */
int hf_modconv3x3_up_blur_f16_f32(float *out, void *split_hi, void *split_lo, const float *x, const void *x_hi,
                                  const void *x_lo, const void *wt_hi, const void *wt_lo, const float *s, const float *d,
                                  const float *blur_k1d_x, const float *blur_k1d_y, const float *noise, const float *noise_w,
                                  long long noise_bstride, const float *bias, const float *s_next, int batch, int cin, int cout,
                                  int h, int w, float alpha, float scale, void *stream);

/* Scratch (in floats) the two modulated-conv entry points need for this shape: the
 * small-plane layers (4x4 .. 16x16) run split-K over the input channels and reduce the
 * partial sums in a second, deterministic pass.  0 = no workspace needed (workspace
 * may then be NULL).  Too small a workspace returns HF_E_WORKSPACE. */
long long hf_modconv_workspace_floats(int batch, int cin, int cout, int h, int w, int upsample);

/* ---------------------------------------------------------------------------
 * Upsampling modulated conv, part 1: per-sample conv_transpose2d(stride 2,
 * pad 0) of s*x with the 3x3 weights, demodulated: tmp[b,co,2h+1,2w+1].
 * Replaces: models/stylegan2/model.py:252-262 (F.conv_transpose2d with
 * groups=batch).  Part 2 is hf_blur_noise_bias_act_f32 below.
 */
int hf_modconv3x3_up_f32(float *tmp, const float *x, const float *wt, const float *s, const float *d,
                         int batch, int cin, int cout, int h, int w, int tmp_pitch, float *workspace,
                         long long workspace_floats, void *stream);
/* Row pitch (floats, multiple of 4 >= 2w+1) of the intermediate: tmp is
 * [batch, cout, 2h+1, pitch]; the padding keeps the blur pass's 16-byte loads aligned. */
int hf_modconv_up_pitch(int w);

/* Part 2: 4x4 FIR blur with pad (1,1) (upfirdn2d mode 1) fused with noise +
 * bias + leaky relu: in [planes=batch*channels, in_h, in_w] -> out
 * [planes, in_h-1, in_w-1].  Replaces: Blur.forward (models/stylegan2/model.py:
 * 77-93, called :263) + NoiseInjection + FusedLeakyReLU.  Rows of `in` are in_pitch floats
 * apart (in_pitch >= in_w; hf_modconv_up_pitch for the fused path).  kernel4x4: device
 * pointer to the module's `blur.kernel` buffer (already multiplied by
 * upsample_factor**2, model.py:83-84).  noise/bias NULL as in hf_modconv3x3_f32. */
int hf_blur_noise_bias_act_f32(float *out, const float *in, const float *kernel4x4, const float *noise,
                               const float *noise_w, long long noise_bstride, const float *bias,
                               int batch, int channels, int in_h, int in_w, int in_pitch, float alpha,
                               float scale, void *stream);

/* hf_blur_noise_bias_act_f32 for a consumer on the fp16 matrix cores: writes
 *   s_next[b,c] * y[b,c,Y,X]   (y = the fp32 result above; s_next NULL = 1)
 * split into fp16 pairs hi = fp16(v), lo = fp16(v - hi) and K-blocked,
 *   out_hi / out_lo [batch][channels/8][out_h][out_w][8]      (channels % 8 == 0),
 * the layout hf_modconv3x3_f16_pre_f32 stages by LDS-DMA.  s_next is the NEXT conv's modulation
 * (hf_modulation_f32 of that layer): the product with the activation has to happen in fp32 before
 * the split.  Bit-identical to splitting the fp32 activation inside the conv kernel.  out_lo may be
 * NULL for a consumer that runs with nterms 1 (plain fp16 operands): half the bytes written. */
int hf_blur_noise_bias_act_split_f16(void *out_hi, void *out_lo, const float *in, const float *kernel4x4,
                                     const float *noise, const float *noise_w, long long noise_bstride,
                                     const float *bias, const float *s_next, int batch, int channels, int in_h,
                                     int in_w, int in_pitch, float alpha, float scale, void *stream);

/* ---------------------------------------------------------------------------
 * ToRGB: 1x1 modulated conv WITHOUT demodulation + bias + upsampled skip:
 *   y[b,c] = sum_ci wt[ci,c] * s[b,ci] * x[b,ci] + bias[c] + upfirdn2d(skip, k4, up=2, pad=(2,1))[b,c]
 * Replaces: ToRGB.forward (models/stylegan2/model.py:356-365) incl.
 * Upsample.forward (:49-53).  wt: [cin, 3] from hf_modconv_prepare_f32 (k=1).
 * skip: [batch,3,h/2,w/2] or NULL; kernel4x4: the `upsample.kernel` buffer.
 */
int hf_torgb_f32(float *out, const float *x, const float *wt, const float *s, const float *bias,
                 const float *skip, const float *kernel4x4, int batch, int cin, int h, int w,
                 void *stream);

/* ===========================================================================
 * Encoders (e4e Encoder4Editing, FeatureStyle fs_encoder_v2): the convolution with its
 * surrounding inference-mode BatchNorm / activation / residual folded in, and the small
 * operators around it.  All tensors NCHW fp32.
 * =========================================================================== */

/* Conv2d weight [cout,cin,k,k] (torch layout) -> wt[k*k][cin][cout] * scale; k = 1, 3 or 7.
 * One-time re-layout of frozen parameters. */
int hf_conv_prepare_f32(float *wt, const float *weight, int cout, int cin, int k, float scale, void *stream);

/* BatchNorm2d with running statistics -> affine: scale = gamma/sqrt(var+eps),
 * shift = beta + (conv_bias - mean)*scale (conv_bias may be NULL).  Replaces the
 * F.batch_norm(..., training=False) calls of helpers.py:100-115 / iresnet.py:45-52. */
int hf_bn_fold_f32(float *scale, float *shift, const float *gamma, const float *beta, const float *mean,
                   const float *var, const float *conv_bias, float eps, int n, void *stream);

/* y = act( out_scale[co] * conv_{k x k, stride, pad k/2}( in_scale[ci]*x + in_shift[ci] ) + bias[co] ) + residual
 *   in_scale / in_shift : the BatchNorm BEFORE the conv (applied to real pixels only, the zero
 *                          padding stays zero - which is why it cannot be folded into the weights);
 *   out_scale / bias    : the BatchNorm AFTER the conv (and/or the conv's own bias);
 *   act                 : 0 none, 1 LeakyReLU(alpha), 2 PReLU(slope[co]), 3 QuickGELU x*sigmoid(1.702x) (CLIP's MLP);
 *   residual            : [batch,cout,oh,ow] added last (IBasicBlock's `out += identity`).
 * Any of the pointers may be NULL.  k in {1,3,7} (7: ungrouped), stride in {1,2}; oh = (h-1)/stride + 1.
 * Replaces: nn.Conv2d + BatchNorm2d + PReLU/LeakyReLU (+ add) chains of
 * helpers.py:99-115, psp_encoders.py:41-47, iresnet.py:44-56, feature_style_encoder.py:33-40.
 * groups > 1: `groups` independent convolutions of this shape in one launch (the e4e style
 * heads, psp_encoders.py:34-55): wt [groups][k*k][cin][cout], bias/out_scale/slope
 * [groups][cout], out/residual [groups][batch][cout][oh][ow]; group g reads
 * x + g*x_group_stride (0 = all groups share one input); in_scale/in_shift must be NULL.
 * workspace: hf_conv2d_workspace_floats() floats (0 -> may be NULL). */
int hf_conv2d_f32(float *out, const float *x, const float *wt, const float *in_scale, const float *in_shift,
                  const float *out_scale, const float *bias, int act, const float *slope, float alpha,
                  const float *residual, int batch, int cin, int cout, int h, int w, int k, int stride,
                  int groups, long long x_group_stride, float *workspace, long long workspace_floats,
                  void *stream);
long long hf_conv2d_workspace_floats(int batch, int cin, int cout, int h, int w, int k, int stride, int groups);

/* hf_conv2d_f32 for k = 3 on the fp16 matrix cores (v_mfma_f32_32x32x16_f16; csrc/convh_enc.hip): same
 * arithmetic contract - in_scale/in_shift on real pixels, out_scale/bias, activation, residual, groups -
 * with the products in the operand modes of hf_modconv3x3_f16_f32 (nterms 3: fp32 operands split into
 * fp16 (hi, lo) pairs, fp32-class accuracy; nterms 1: operands rounded to fp16), fp32 tensors and
 * accumulation.  Replaces the same reference chains as hf_conv2d_f32 (helpers.py:99-115,
 * psp_encoders.py:41-47, iresnet.py:44-56, feature_style_encoder.py:33-40, models/Encoders.py:35-57).
 * wt_hi / wt_lo: per group, hf_conv_split_weights_f16 of that group's prepared weights
 * ([9][cin][cout] from hf_conv_prepare_f32): group g's hi block starts at wt_hi + g*(9*cin*cout + 8) halves
 * (its 16-byte trailer included), its lo block at wt_lo + g*9*cin*cout halves.
 * Shapes: cin % 16 == 0, cout % 64 == 0, output planes at least 9 pixels wide and tall enough for one
 * 128-pixel tile of 16 or 32 columns; anything else returns HF_E_INVALID and the caller uses
 * hf_conv2d_f32.
 * x_hi / x_lo (NULL = off): the input already transformed, split and K-blocked by
 * hf_split_activation_f16 ([images][cin/8][h][w][8] fp16; x, in_scale, in_shift then unused and the
 * latter two must be NULL): the kernel stages activations by LDS-DMA - worth it when one input feeds
 * many output-channel tiles / groups (the e4e style heads: 88 block columns per input tile).  With
 * groups > 1 and x_group_stride != 0 the split tensors hold [groups][batch] images.
 * workspace: hf_conv2d_f16_workspace_floats() floats (layers whose output grid cannot fill the chip
 * run split-K over the input channels + a deterministic second pass); 0 -> may be NULL. */
int hf_conv2d_f16_f32(float *out, const float *x, const void *x_hi, const void *x_lo, const void *wt_hi,
                      const void *wt_lo, int nterms, const float *in_scale, const float *in_shift,
                      const float *out_scale, const float *bias, int act, const float *slope, float alpha,
                      const float *residual, int batch, int cin, int cout, int h, int w, int stride, int groups,
                      long long x_group_stride, float *workspace, long long workspace_floats, void *stream);
long long hf_conv2d_f16_workspace_floats(int batch, int cin, int cout, int h, int w, int stride, int groups);
/* hf_conv2d_f16_f32 (groups = 1) whose result y also - or, out == NULL, only - leaves as the PRE-SPLIT input of the next
 * fp16-core convolution: next_scale[co] * y + next_shift[co] (the consumer's input affine, e.g. its folded BatchNorm;
 * NULL = identity) as fp16 (hi, lo) parts in the K-blocked layout of hf_split_activation_f16, written by the conv's own
 * epilogue - the conv1 -> PReLU -> conv2 hand-off inside an IR-SE / IBasicBlock unit (helpers.py:99-115, iresnet.py:44-56)
 * then costs no fp32 round trip and no split pass.  out_lo may be NULL for nterms 1.  Refused (HF_E_INVALID) when the
 * shape would run split-K (hf_conv2d_f16_workspace_floats(...) != 0): the caller then uses the two-call form. */
int hf_conv2d_f16_split_f32(float *out, void *out_hi, void *out_lo, const float *next_scale, const float *next_shift,
                            const float *x, const void *x_hi, const void *x_lo, const void *wt_hi, const void *wt_lo,
                            int nterms, const float *in_scale, const float *in_shift, const float *out_scale,
                            const float *bias, int act, const float *slope, float alpha, const float *residual, int batch,
                            int cin, int cout, int h, int w, int stride, void *stream);
/* hf_scale_shortcut_add_f32 (the tail of bottleneck_IR_SE.forward, helpers.py:118-120: res * gate + shortcut, shortcut optionally
 * MaxPool2d(1, stride), :95-96) whose result ALSO leaves as the pre-split input of the next unit's first 3x3 conv (ABI 11):
 * out = r * gate + shortcut [batch, channels, oh, ow] fp32, and next_scale[c] * out + next_shift[c] (the next unit's BatchNorm,
 * helpers.py:99; NULL = identity) as fp16 (hi, lo; out_lo NULL = plain fp16 consumer) in hf_split_activation_f16's layout - bit for
 * bit what the two separate passes write.  channels % 8 == 0. */
int hf_scale_shortcut_add_split_f16(float *out, void *out_hi, void *out_lo, const float *next_scale, const float *next_shift,
                                    const float *r, const float *gate, const float *shortcut, int sc_stride, int batch, int channels,
                                    int oh, int ow, int sh, int sw, void *stream);
/* 1 when hf_conv2d_f16_split_f32 takes a launch of this shape (ABI 11): the split output is written by the conv kernel's own
 * epilogue, so a launch that spreads its K loop over the grid (hf_conv2d_f16_workspace_floats != 0 for it) cannot produce it.
 * In batch-invariant mode a launch that fills the chip by itself runs its K partition inside the blocks instead (same bits),
 * which the register-staged 512-pixel tile form has no registers for: the answer depends on `presplit_input`
 * (x_hi / x_lo given) and `nterms`. */
int hf_conv2d_f16_split_output_ok(int batch, int cin, int cout, int h, int w, int stride, int nterms, int presplit_input);
/* hf_conv2d_f32 for k = 1 on the fp16 matrix cores (csrc/gemm_h.hip): a GEMM over the pixels,
 *   y = act( out_scale[co] * sum_ci W[co,ci] * (in_scale[ci]*x + in_shift[ci]) + bias[co] ) + residual,
 * stride in {1, 2} (the source pixel of output (oy, ox) is (oy*stride, ox*stride)), operand modes as hf_conv2d_f16_f32.
 * Replaces the 1x1 shortcut / downsample convolutions of the encoders (helpers.py:99-103, iresnet.py:17-19), BiSeNet's
 * and SEAN's 1x1 convs (architecture.py:50), and - on feature-major activations x[feature][token], i.e. NCHW with the
 * tokens as pixels - the nn.Linear layers of the CLIP image tower (clip/model.py) and SEAN's per-label table GEMM.
 * wt_hi / wt_lo: hf_conv_split_weights_f16_taps(taps = 1) of the prepared [1][cin][cout] weights; per group a
 * self-contained [cin*cout halves | 16-byte trailer] (hi) and cin*cout halves (lo).  cin % 32 == 0, cout % 64 == 0,
 * otherwise HF_E_INVALID (callers use hf_conv2d_f32).  groups > 1 as in hf_conv2d_f32 (in_scale / in_shift NULL).
 * x_hi / x_lo (NULL = off; stride 1, groups 1, no in_scale / in_shift): the input pre-split by hf_split_activation_f16
 * ([images][cin/8][h*w][8] fp16), staged by LDS-DMA - worth it when one input feeds many 64-channel output tiles.
 * workspace: hf_conv1x1_f16_workspace_floats() floats (small grids split K; 0 -> may be NULL). */
int hf_conv1x1_f16_f32(float *out, const float *x, const void *x_hi, const void *x_lo, const void *wt_hi,
                       const void *wt_lo, int nterms, const float *in_scale, const float *in_shift, const float *out_scale, const float *bias, int act,
                       const float *slope, float alpha, const float *residual, int batch, int cin, int cout, int h, int w,
                       int stride, int groups, long long x_group_stride, float *workspace, long long workspace_floats,
                       void *stream);
long long hf_conv1x1_f16_workspace_floats(int batch, int cin, int cout, int h, int w, int stride, int groups);
/* Small-plane modulated 3x3 convolution (h*w <= 1024; the generator's 4^2 .. 16^2 layers) on the fp16 matrix cores:
 * the nine taps as one 1x1 GEMM with 9*cout output rows over the modulated input (no halo, no shifts), then a combine
 * pass that adds the taps at their shifted positions - and the split-K partials - in a fixed order and applies the tail.
 *   upsample == 0: hf_modconv3x3_f32's contract (ModulatedConv2d.forward same-resolution branch, model.py:238-250,
 *     273-277, + noise / bias / leaky ReLU of StyledConv.forward :337-343): out [batch,cout,h,w];
 *   upsample != 0: hf_modconv3x3_up_f32's (F.conv_transpose2d stride 2, model.py:252-262): out = the demodulated
 *     [batch,cout,2h+1,tmp_pitch] intermediate for hf_blur_noise_bias_act_*; noise / bias must be NULL.
 * w9_hi / w9_lo: hf_conv_split_weights_f16_taps(taps = 1, cin, 9*cout) of the prepared weights re-laid out
 * [1][cin][tap*cout + co] (from hf_modconv_prepare_f32's wt[tap][ci][co]).  cin % 32 == 0, cout % 64 == 0.
 * workspace: hf_modconv3x3_small_workspace_floats() floats (the per-tap products, one slab per K split). */
int hf_modconv3x3_small_f16_f32(float *out, const float *x, const void *w9_hi, const void *w9_lo, int nterms, const float *s,
                                const float *d, const float *noise, const float *noise_w, long long noise_bstride,
                                const float *bias, int batch, int cin, int cout, int h, int w, float alpha, float scale,
                                int upsample, int tmp_pitch, float *workspace, long long workspace_floats, void *stream);
long long hf_modconv3x3_small_workspace_floats(int batch, int cin, int cout, int h, int w);
/* The small-plane UPSAMPLING StyledConv in two launches (round 6, ABI 12): hf_modconv3x3_small_f16_f32(upsample = 1) followed by
 * hf_blur_noise_bias_act_f32 (out != NULL) or hf_blur_noise_bias_act_split_f16 (out_hi != NULL; out_lo NULL for an nterms-1
 * consumer) - F.conv_transpose2d stride 2, Blur, NoiseInjection, FusedLeakyReLU of the upsampling StyledConv
 * (models/stylegan2/model.py:252-263, :337-343) - with the (2h+1)^2 intermediate kept in LDS: after the tap GEMM one kernel per
 * (image, 8-channel block) combines the taps, blurs and applies the tail.  Exactly one of out [batch,cout,2h,2w] / out_hi;
 * s_next [batch,cout] (NULL = 1): the consumer's modulation, multiplied in before the split.  blur_kernel4x4: the module's
 * [4][4] blur buffer (applied flipped: a true convolution, op/upfirdn2d.py:186).  Results equal the three-launch path bit for
 * bit.  (8*(2h+3)*(2w+3) + 72*h*w) floats of LDS <= 150 KB (inputs up to 16 x 16 + a little); workspace as hf_modconv3x3_small_f16_f32. */
int hf_modconv3x3_small_up_blur_f16_f32(float *out, void *out_hi, void *out_lo, const float *x, const void *w9_hi,
                                        const void *w9_lo, int nterms, const float *s, const float *d,
                                        const float *blur_kernel4x4, const float *noise, const float *noise_w,
                                        long long noise_bstride, const float *bias, const float *s_next, int batch, int cin,
                                        int cout, int h, int w, float alpha, float scale, float *workspace,
                                        long long workspace_floats, void *stream);
/* in_scale[c]*x + in_shift[c] (NULL = identity) split into fp16 pairs hi = fp16(v), lo = fp16(v - hi)
 * (saturating, hf_f16_overflow_count) and K-blocked: out_hi / out_lo [images][channels/8][h][w][8];
 * x [images][channels][h][w] fp32, channels % 8 == 0.  out_lo may be NULL (nterms 1 consumer).
 * The producer-side form of the conversion hf_conv2d_f16_f32 otherwise does per block while staging. */
int hf_split_activation_f16(void *out_hi, void *out_lo, const float *x, const float *in_scale, const float *in_shift,
                            long long images, int channels, int h, int w, void *stream);

/* hf_split_activation_f16 with a per-IMAGE scale: scale[image][channel] * x (NULL = 1) - the modulation s of a
 * ModulatedConv2d applied to its input (models/stylegan2/model.py:241-248) - for the pre-split consumers
 * (hf_conv1x1_f16_f32's x_hi / x_lo; used inside hf_modconv3x3_small_f16_f32). */
int hf_split_activation_mod_f16(void *out_hi, void *out_lo, const float *x, const float *scale, long long images,
                                int channels, int h, int w, void *stream);

/* out[p] = mean of plane p (AdaptiveAvgPool2d(1) of SEModule, helpers.py:60,68). */
int hf_plane_mean_f32(float *out, const float *x, int planes, int hw, void *stream);
/* gate[b,c] = sigmoid(fc2 . relu(fc1 . pooled[b])) ; fc1 [reduced,channels], fc2 [channels,reduced]
 * (the two bias-free 1x1 convs of SEModule, helpers.py:61-73). */
int hf_se_gate_f32(float *gate, const float *pooled, const float *fc1, const float *fc2, int batch, int channels,
                   int reduced, void *stream);
/* out = r * gate[b,c] + shortcut[b,c, y*sc_stride, x*sc_stride]; gate may be NULL.
 * SEModule's `module_input * x` + bottleneck_IR_SE's `res + shortcut` with the
 * MaxPool2d(1, stride) shortcut (helpers.py:73, 95-96, 118-120). */
int hf_scale_shortcut_add_f32(float *out, const float *r, const float *gate, const float *shortcut,
                              int sc_stride, int batch, int channels, int oh, int ow, int sh, int sw,
                              void *stream);
/* out = bilinear_resize(x [planes,h,w] -> [oh,ow], align_corners=True) + y  (_upsample_add, helpers.py:123-140) */
int hf_upsample_bilinear_add_f32(float *out, const float *x, const float *y, int planes, int h, int w, int oh,
                                 int ow, void *stream);
/* AdaptiveAvgPool2d((oh,ow)) of x [batch,channels,h,w], written into channels
 * [offset, offset+channels) of out [batch,out_channels_total,oh,ow] (pool + torch.cat of
 * feature_style_encoder.py:52-61). */
int hf_adaptive_avgpool_f32(float *out, const float *x, int batch, int channels, int h, int w, int oh, int ow,
                            int out_channels_total, int out_channel_offset, void *stream);
/* F.interpolate(x, scale_factor=0.5, mode='bilinear') on [planes,h,w], h and w even (trainer.py:61-64). */
int hf_downscale2x_f32(float *out, const float *x, int planes, int h, int w, void *stream);
/* out[b,n] = scale * sum_k x[b*x_stride+k] * w[n*in_features+k] + bias[n] (rows in chunks of 8 per block column).
 * nn.Linear (scale 1; feature_style_encoder.py:46, 62-63) and EqualLinear (scale
 * 1/sqrt(in_features); psp_encoders.py:48, 53). */
int hf_linear_f32(float *out, const float *x, long long x_stride, const float *w, const float *bias, int batch,
                  int in_features, int out_features, float scale, void *stream);
/* EqualLinear.forward (models/stylegan2/model.py:153-163) as one launch:
 *   out[b,n] = (sum_k x[b,k] * w[n,k]) * lr_mul/sqrt(in_features) + bias[n] * lr_mul
 * and, with fused_lrelu != 0 (activation='fused_lrelu', :154-156), leaky_relu(., alpha) * act_scale on
 * top - one layer of the z -> w mapping network (:384-393: lr_mul 0.01, alpha 0.2, act_scale sqrt 2).
 * bias may be NULL. */
int hf_equal_linear_f32(float *out, const float *x, long long x_stride, const float *w, const float *bias, int batch,
                        int in_features, int out_features, float lr_mul, int fused_lrelu, float alpha, float act_scale,
                        void *stream);
/* PixelNorm.forward (models/stylegan2/model.py:16-21) on [rows, dim]: x * rsqrt(mean_k x^2 + 1e-8). */
int hf_pixel_norm_f32(float *out, const float *x, int rows, int dim, void *stream);
/* ---- BiSeNet face parsing + get_segmentation (models/CtrlHair/external_code/face_parsing/model.py:230-253,
 * resnet.py:19-77, my_parsing_util.py:72-95, models/Net.py:108-115) - its convolutions are hf_conv2d_f32 /
 * hf_conv2d_f16_f32 (k = 7 for the ResNet stem; act | HF_ACT_RESIDUAL_FIRST for `relu(shortcut + residual)`) ---- */
/* nn.MaxPool2d(kernel_size=3, stride=2, padding=1) on [planes, h, w] -> [planes, (h-1)/2+1, (w-1)/2+1] (resnet.py:62) */
int hf_maxpool3x3s2_f32(float *out, const float *x, long long planes, int h, int w, void *stream);
/* out[p][i] = x[p][i] * (sigmoid(logit[p]) + plus_one) + add_plane[p][i] + add_bcast[p]   (add_* may be NULL):
 * AttentionRefinementModule's `feat * sigmoid(atten)` with the following `+ avg_up` / `+ feat32_up` (model.py:80-86,
 * 111-122; avg_up is a 1x1 map broadcast over the plane), FeatureFusionModule's `feat * atten + feat` (plus_one = 1). */
int hf_gate_f32(float *out, const float *x, const float *logit, const float *add_plane, const float *add_bcast,
                float plus_one, long long planes, int hw, void *stream);
/* F.interpolate(mode='nearest') of [planes, h, w] to [planes, oh, ow] (model.py:112, 116, 121) */
int hf_upsample_nearest_f32(float *out, const float *x, long long planes, int h, int w, int oh, int ow, void *stream);
/* The tail of get_segmentation in one pass: bilinear (align_corners=True) up-sampling of the class logits
 * [images, classes, h, w] to (full_h, full_w) (model.py:249) evaluated ONLY at the source pixels of the nearest resize to
 * (out_h, out_w) (models/Net.py:112-114; out = full for resize=False), argmax over the classes (first maximum,
 * my_parsing_util.py:86) and the label permutation remap[classes] (swap_parsing_label_to_celeba_mask, :89-95; NULL = none).
 * The interpolation follows ATen's formula and operation order without fused multiply-adds: equal logits give equal
 * indices.  out: int64 [images, out_h, out_w]. */
int hf_parsing_mask_i64(long long *out, const float *logits, const int *remap, int images, int classes, int h, int w,
                        int full_h, int full_w, int out_h, int out_w, void *stream);

/* ---- stencils on either side of the hot path ---- */
/* LayerNorm of the CtrlHair shape adaptor's Conv2dBlock (models/CtrlHair/my_torchlib/module.py:181-206, norm 'ln'):
 * per sample, over all channels * hw values: y = (x - mean) / (std_unbiased + eps) * gamma[c] + beta[c], then
 * LeakyReLU(slope) (slope 1 = no activation; gamma / beta may be NULL).  out: [batch][channels][hw]; x: the same, or
 * with x_batch_stride (> channels*hw; 0 = dense) elements between samples - the first `channels` planes of a tensor
 * with more (a conv output whose channel count was padded up to a tile multiple). */
int hf_sample_layernorm_f32(float *out, const float *x, const float *gamma, const float *beta, int batch, int channels,
                            int hw, long long x_batch_stride, float eps, float slope, float *workspace,
                            long long workspace_floats, void *stream);
/* scratch of hf_sample_layernorm_f32 (per-chunk statistics), in floats */
long long hf_sample_layernorm_workspace_floats(int batch, int channels, int hw);
/* BicubicDownSample.forward (utils/bicubic.py:38-75): reflect padding + the separable 4*factor-tap bicubic filter
 * k1d (device, 4*factor floats, normalised; the module's k), stride factor along both axes:
 * x [planes, h, w] -> out [planes, h/factor, w/factor]; h, w multiples of factor (the pipeline: 1024 -> 512 / 256). */
int hf_bicubic_down_f32(float *out, const float *x, const float *k1d, long long planes, int h, int w, int factor,
                        void *stream);
/* DilateErosion.mask (utils/image_utils.py:42-55) on a BINARY (0/1) mask [planes, h, w]: `radius` = dilate_erosion
 * rounds of the 4-neighbourhood cross, i.e. one pass with the diamond |dy|+|dx| <= radius; pixels outside the image
 * count as unset (the conv's zero padding).  dilated / eroded: 0/1 floats, same shape. */
int hf_dilate_erode_f32(float *dilated, float *eroded, const float *mask, long long planes, int h, int w, int radius,
                        void *stream);

/* ---- PostProcessModel's latent branch (models/Encoders.py:13-32, 119-131) ----
 * F.layer_norm over the last `dim` elements of each of `rows` rows (biased variance, eps inside the sqrt):
 * gamma / beta [dim] = elementwise affine (both NULL: LayerNorm(elementwise_affine=False), :19), lrelu != 0
 * applies LeakyReLU(alpha) on top (the LayerNorm -> LeakyReLU of gamma_function / beta_function, :20-21). */
int hf_layernorm_f32(float *out, const float *x, const float *gamma, const float *beta, int rows, int dim, float eps,
                     int lrelu, float alpha, void *stream);
/* hf_layernorm_f32 for `groups` LayerNorms at once: x is [rows][dim] with rows % groups == 0, gamma / beta are
 * [groups][dim] and row r takes group r % groups - the LayerNorm + LeakyReLU of the 2 x 5 gamma / beta branches of a
 * ModulationModule stack (models/Encoders.py:20-23) over the columns of ONE stacked first Linear, viewed as
 * [rows * groups, dim].  Same per-row arithmetic as hf_layernorm_f32. */
int hf_layernorm_grouped_f32(float *out, const float *x, const float *gamma, const float *beta, int rows, int dim, int groups,
                             float eps, int lrelu, float alpha, void *stream);
/* out = x * (1 + gamma) + beta (all [n]), lrelu != 0: LeakyReLU(alpha) on top (ModulationModule.forward :29-31). */
int hf_modulate_f32(float *out, const float *x, const float *gamma, const float *beta, long long n, int lrelu, float alpha,
                    void *stream);
/* PixelNorm over dim 1 of x [batch, layers, dim] - what models/stylegan2/model.py:16-21 computes when the
 * reference applies it to W+ codes (Encoders.py:123-124): x * rsqrt(mean over the layers of x^2 + 1e-8). */
int hf_pixel_norm_dim1_f32(float *out, const float *x, int batch, int layers, int dim, void *stream);
/* out[i] = alpha * a[i] + beta * b[i % b_period]   (latent_avg + 0.1 * (dt_face + dt_hair), Encoders.py:131) */
int hf_axpby_bcast_f32(float *out, const float *a, float alpha, const float *b, float beta, long long n, long long b_period,
                       void *stream);
/* out[i] = a[i] + b[i % b_period] */
int hf_add_bcast_f32(float *out, const float *a, const float *b, long long n, long long b_period, void *stream);

/* ===========================================================================
 * SEAN inpainting stage: encode_sean / decode_sean of
 * models/sean_codes/models/pix2pix_model.py:299-325 around SPADEGenerator (networks/generator.py:72-110),
 * SPADEResnetBlock / Zencoder (networks/architecture.py) and ACE / SPADE (networks/normalization.py).  Its dense 3x3
 * convolutions are hf_conv2d_f32 / hf_conv2d_f16_f32; the entry points below are the label-driven parts.  csrc/sean.hip.
 * =========================================================================== */
/* 3x3 convolution (zero pad 1) of an input that is constant per segmentation label, as a table lookup:
 *   out[b,c,y,x] = act( bias[c] + sum_{ky,kx} table[((ky*3+kx)*channels + c) * table_cols
 *                                                   + b*cols_per_sample + labels[b/group, y+ky-1, x+kx-1]] )
 * table [9*channels][table_cols] holds W[c, :, ky, kx] . v_label for every label's input vector v_label:
 *   - one-hot label maps (SPADE's mlp_shared, normalization.py:246-252; SPADEGenerator.fc, generator.py:75-76):
 *     table = the conv weight itself re-laid out, cols_per_sample = 0;
 *   - ACE's conv_gamma / conv_beta on `middle_avg` (normalization.py:117-162: relu(fc_mu_j(style code j)) broadcast
 *     over region j): table = a [19*batch]-column GEMM of the weights with the per-sample region vectors,
 *     cols_per_sample = 19.
 * labels: int32 [batch/group, h, w] (group consecutive samples share a label map - the two decodes of a pair,
 * models/Alignment.py:130-131); relu != 0 applies ReLU; bias may be NULL.
 * tap_sum_scratch: NULL, or channels*table_cols floats of scratch: non-NULL turns on the interior path - the sum of the
 * nine taps is formed first (0 + tap 0 + ... + tap 8) and pixels whose whole 3x3 neighbourhood carries one label (the
 * interior of a region) take ONE lookup instead of nine; worth it from ~32x32 planes upward.  Planes of >= 1024 pixels
 * with w % 4 == 0 and <= 32 labels per sample keep their tables in LDS, four pixels per thread, and choose interior /
 * general per PIXEL (the scratch is then not written); smaller ones choose per wave of 64 pixels. */
int hf_label_conv3x3_f32(float *out, const int *labels, const float *table, const float *bias, int batch, int channels,
                         int h, int w, int table_cols, int cols_per_sample, int group, int relu, float *tap_sum_scratch,
                         void *stream);
/* The tail of ACE.forward (normalization.py:103-107, 164-185) in one pass:
 *   n   = (x + noise[b,p] * noise_var[c]) * bn_scale[c] + bn_shift[c]      (the eval-mode SynchronizedBatchNorm2d as an
 *                                                                           affine: hf_bn_fold_f32 with gamma 1, beta 0)
 *   g   = sigmoid(blend[0]) * avg[b,c]   + (1 - sigmoid(blend[0])) * sp[b/group, c]
 *   bt  = sigmoid(blend[1]) * avg[b,C+c] + (1 - sigmoid(blend[1])) * sp[b/group, C+c]     (avg NULL: g, bt = sp's)
 *   out = LeakyReLU_slope( n * (1 + g) + bt )                                (slope 1 = none; SPADEResnetBlock.actvn)
 * x, out [batch,channels,hw]; noise [batch,hw] or NULL (ACE draws randn(B,W,H,1) and transposes: the caller's layout
 * choice); avg [batch,2*channels,hw], sp [batch/group,2*channels,hw]: gamma planes then beta planes; blend: DEVICE
 * pointer to (blending_gamma, blending_beta); hw % 4 == 0.
 * x_upsample_w != 0 (ABI 9): x is the HALF-resolution tensor [batch,channels,h/2,w/2] with w = x_upsample_w and the
 * generator's nearest-neighbour x2 up-sampling (`self.up`, generator.py:80-103) is read in place - x[.., y/2, x/2] - instead
 * of being materialised; w % 4 == 0, h = hw / w even. */
int hf_ace_modulate_f32(float *out, const float *x, const float *noise, const float *noise_var, const float *bn_scale,
                        const float *bn_shift, const float *avg, const float *sp, const float *blend, int batch,
                        int channels, int hw, int group, float slope, int x_upsample_w, void *stream);
/* hf_label_conv3x3_f32 (cols_per_sample = n_labels, bias = avg_bias, channels = 2*channels, no ReLU) FOLLOWED BY
 * hf_ace_modulate_f32 with its result as `avg`, in one pass and bit-identical to the pair: the avg planes - ACE's
 * conv_gamma / conv_beta of `middle_avg` (normalization.py:117-162) - are looked up from LDS copies of the table rows
 * instead of being written to and re-read from HBM.  table [9 * 2*channels][table_cols] (rows (tap, gamma|beta
 * channel)), columns b*n_labels + label; labels int32 [batch/group, h, w]; sp [batch/group, 2*channels, h, w]; x / out
 * [batch, channels, h, w]; noise [batch, h*w] or NULL; avg_bias [2*channels] or NULL; blend as above (required);
 * interior != 0: regions' interior pixels take the tap-sum path; w % 4 == 0, n_labels <= 32.  x_upsample != 0: x is
 * [batch, channels, h/2, w/2], up-sampled x2 (nearest) in place as in hf_ace_modulate_f32. */
int hf_ace_modulate_table_f32(float *out, const float *x, const float *noise, const float *noise_var, const float *bn_scale,
                              const float *bn_shift, const int *labels, const float *table, const float *avg_bias,
                              const float *sp, const float *blend, int batch, int channels, int h, int w, int table_cols,
                              int n_labels, int group, float slope, int interior, int x_upsample, void *stream);
/* Zencoder's per-region average pooling (architecture.py:187-205): out[b,l,c] = mean over {p : labels[b,p] == l} of
 * act(x[b,c,p]), 0 for labels that do not occur; act 1 = tanh (the encoder's last layer, :177), 0 = none.
 * x is a strided view: sample stride batch_stride, plane stride plane_stride, row pitch `pitch` (floats).
 * labels int32 [batch,h,w]; n_labels must be 19.  out [batch,19,channels]. */
int hf_region_mean_f32(float *out, const float *x, const int *labels, int batch, int channels, int h, int w, int n_labels,
                       long long batch_stride, long long plane_stride, int pitch, int act, void *stream);
/* out = tanh(x), n elements (SPADEGenerator.forward's last line, generator.py:109). */
int hf_tanh_f32(float *out, const float *x, long long n, void *stream);

/* The ResNet-18 stem of BiSeNet's context path (models/CtrlHair/external_code/face_parsing/resnet.py:57-62, 81-84) on the
 * fp16 matrix cores (csrc/stem.hip):  y = LeakyReLU_alpha( conv7x7_stride2_pad3(x) * out_scale[co] + bias[co] )  - conv1,
 * the folded bn1 and ReLU (alpha 0) - and, pool != 0, MaxPool2d(3, 2, 1) of y in the same pass (y is never written).
 * x [batch,3,h,w] fp32; out [batch,cout,oh,ow] with oh = (h-1)/2+1 (pool == 0) or [batch,cout,(oh-1)/2+1,(ow-1)/2+1];
 * cout % 64 == 0; 0 <= alpha <= 1; out_scale / bias [cout] or NULL.
 * Weights: w_hi / w_lo = fp16 (hi, lo) parts of W * 2^k laid out [cout/64][22][64][8 halves] - K group g < 21 = (ci, ky) =
 * (g / 7, g % 7), the eight halves = kx 0..6 and a zero, group 21 all zero - and w_unscale -> 2^-k on the device
 * (hairfastgan_amd/_marshal.py stem_prepare builds them from the [cout,3,7,7] tensor).  Operands are split f16x3
 * (fp32-class, see hf_conv2d_f16_f32). */
int hf_stem7x7s2_f16_f32(float *out, const float *x, const void *w_hi, const void *w_lo, const float *w_unscale,
                         const float *out_scale, const float *bias, float alpha, int batch, int h, int w, int cout, int pool,
                         void *stream);

/* ===========================================================================
 * CLIP ViT-B/32 image tower: `self.clip_model.encode_image(...)` of ClipBlendingModel
 * (models/Encoders.py:75-90; the un-vendored dependency `clip @ git+https://github.com/openai/CLIP@a1d0717`,
 * requirements.txt:6 - clip/model.py VisionTransformer, ResidualAttentionBlock, QuickGELU, LayerNorm).  The linear
 * layers run as 1x1-conv GEMMs (hf_conv2d_f32, k = 1) on FEATURE-MAJOR activations x[feature][token]; the entry
 * points below are the operators between them, on the same layout.  csrc/vit.hip.
 * =========================================================================== */
/* LayerNorm over the feature axis of x [channels][tokens] (per token: biased variance, eps inside the sqrt), affine
 * gamma / beta [channels] (both may be NULL).  clip/model.py LayerNorm (an nn.LayerNorm evaluated in fp32). */
int hf_channel_layernorm_f32(float *out, const float *x, const float *gamma, const float *beta, int channels,
                             long long tokens, float eps, void *stream);
/* Self-attention core of nn.MultiheadAttention (ResidualAttentionBlock.attention, no mask) for short sequences:
 * qkv [3*E][T] = the in_proj output, feature-major (rows 0..E-1 q, E..2E-1 k, 2E..3E-1 v), E = heads*head_dim,
 * T = images*seq (image-major); out [E][T]: per (image, head) softmax(q k^T / sqrt(head_dim)) v, heads concatenated
 * along the feature axis (the input of out_proj).  seq <= 64, head_dim == 64. */
int hf_mha_small_f32(float *out, const float *qkv, int images, int seq, int heads, int head_dim, void *stream);
/* QuickGELU: out = x * sigmoid(1.702 x) (clip/model.py QuickGELU), n elements. */
int hf_quick_gelu_f32(float *out, const float *x, long long n, void *stream);

/* ---------------------------------------------------------------------------
 * Tuning / debugging hook (no reference counterpart): force the tile
 * configuration hf_modconv3x3_f32 / hf_modconv3x3_up_f32 dispatch to
 * (see the switch statements at the bottom of csrc/modconv.hip); 0 restores the
 * built-in heuristics.  Shapes a forced configuration cannot handle fall back
 * to the general kernel.  The setting - like everything the three hf_debug_* hooks touch - is PER CALLING
 * THREAD (thread_local): the library holds no process-global mutable state and its entry points are
 * thread-compatible.  Benchmarks and tests only.
 */
int hf_debug_set_dispatch(int same_cfg, int up_cfg);
/* Which kernel the last modulated-conv call used: 100 * family + tile configuration id,
 * family 1 = general, 2 = pipelined (double-buffered DMA), 3 = split-K (id 0), 5 = fp16 matrix
 * cores (hf_modconv3x3_f16_f32: ids 51-56, 51/52 can be forced through same_cfg; +20 = pre-split input;
 * hf_modconv3x3_up_f16_f32: ids 61/63, +20 = pre-split input; hf_modconv3x3_up_blur_f16_f32: 73, pre-split 93),
 * 6 = hf_conv2d_f16_f32 (1: 64x256 tile,
 * 2: stride 2, 3: 64x128 tile).  Tests use
 * it to make sure a shape exercises the path it is meant to; bench.py to label launches. */
int hf_debug_last_path(void);
/* The fp16 matrix-core kernels launch one resident block per CU and let it walk several tiles
 * as one software pipeline; `blocks` overrides the resident-block count the grid is sized for
 * (0 = the MI355X's 256 CUs).  Tests use small values to exercise the tile hand-over.  Round 5: hf_conv2d_f16_f32's tile forms
 * that are alone on a CU walk their grid the same way (csrc/convh_enc.hip, ConvParams::persist) and size it by this count too. */
int hf_debug_set_persistent_blocks(int blocks);
/* Tuning switches of the fp16 matrix-core kernels (per thread, like the other debug hooks): bit 0 = issue every
 * stage's LDS-DMA copies in the stage's first tap-step instead of spreading them one per tap-step (the default,
 * measured 0-8 % faster on every generator layer); bit 1 = hf_conv1x1_f16_f32 never uses its 128-channel block form (round 6: eight
 * waves sharing one activation stage - taken when the launch fills two rounds of CUs with it; a tile form: equal bits) and the
 * small-plane tap GEMM never trades its 256-pixel tiles for 128-pixel ones (taken when the last round of the 256-pixel form
 * would be under half full: three instead of two blocks per CU; equal bits);
 * bit 2 = hf_conv2d_f16_f32 never uses its 512-pixel tile form,
 * bits 8-15 = the minimum number of 512-pixel blocks / 8 for that form (0 = the default, 512), bits 16-23 = the
 * same for the 256-pixel form (default 384), bit 3 = the fp16-core conv kernels launch their grid columns-fastest whenever that is
 * legal (by default only when it moves fewer bytes from beyond L2; results do not depend on the block order; bits 5-7 are
 * timing ablations of the row-pipeline kernel), bits 24-31 = the block count from which a launch counts as filling the
 * chip by itself (0 = the default, 256; batch-invariant plans: such a launch runs its K partition inside its blocks instead
 * of spreading it over the grid - tests reach that form on small shapes with it).  Results of the generator
 * kernels do not depend on it; tile forms of hf_conv2d_f16_f32 differ in summation order only. */
int hf_debug_set_tuning(int bits);
/* Batch-invariant plans (process-wide; returns the previous setting; set it while no other thread is launching).  Split-K factors and tile forms are normally chosen
 * from the whole launch (batch x pixels): the chip is filled, but a sample's summation order - and with it the last bits of
 * its result - depends on what it is batched with.  on != 0: every plan is made from the per-sample shape only, so that a
 * sample gives the same bits at any batch size; the reference's north_star asks for bit-exact segmentation-mask indices,
 * and an argmax near a tie must not flip between `swap` and `swap_batch` (models/.../my_parsing_util.py:85-86,
 * shape_branch/solver.py:248-262).  Workspace queries answer for the current setting.  Cost: batched launches keep the
 * split-K passes and the small tile forms of a batch-1 launch. */
int hf_set_batch_invariant(int on);
/* In-kernel second half of split-K launches (per thread).  Small launches split their K loop over blockIdx.z
 * (hf_conv2d_f32 / hf_modconv3x3_f32 on small planes, hf_conv2d_f16_f32, hf_conv1x1_f16_f32); by default a second
 * kernel (splitk_reduce) adds the z slabs and runs the epilogue.  With a registered buffer of `n_ints` ZEROED 32-bit
 * counters in device memory - caller-owned like every other buffer, one per stream that launches concurrently - the
 * blocks of an output tile count their arrivals instead and the LAST one adds the slabs (in z order: the same bits as
 * the second kernel, whichever block is last) and finishes the tile; every launch leaves the buffer zero.  Launches with
 * more output tiles than counters, and the transposed forms, keep the two-kernel form.  NULL unregisters. */
int hf_set_splitk_counters(void *zeroed_ints, int n_ints);
/* Profiling aid (ABI 11; no reference counterpart - the reference's harness is a wall-clock decorator, utils/time.py:9-36):
 * launches the empty kernel `hf_profile_marker_kernel` with id + 1 workgroups (0 <= id <= 1023) on `stream`.  Placed
 * around a region, it lets rocprofv3's per-dispatch tables be cut to that region (tools/summarize_prof.py --between). */
int hf_profile_marker(int id, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HAIRFAST_HIP_H */
