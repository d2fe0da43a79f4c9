// vit.hip - the small operators of the CLIP ViT-B/32 image tower (models/Encoders.py:75-90: `clip_model.encode_image`;
// OpenAI CLIP clip/model.py VisionTransformer / ResidualAttentionBlock / QuickGELU / LayerNorm).  The linear layers are
// the library's 1x1-conv GEMMs on FEATURE-MAJOR activations x[feature][token] (an NCHW tensor whose "pixels" are the
// tokens); these kernels work on the same layout so that nothing is transposed between layers.
#include "hf_common.h"

// Floating-point contraction: "on" = a multiply and an add are fused only where they are written in ONE expression (or as
// fmaf), never across statements.  hipcc's default ("fast") lets the backend fuse opportunistically per basic block: the
// unrolled body of a grid-stride loop and its remainder iterations then round differently, i.e. a sample's bits depend on
// how many elements the launch has - on what it is batched with (found by tools/probes/batch_variance.py in
// upsample_bilinear_add: the third of three images differed from the third of six by one ulp).
#pragma clang fp contract(on)

// LayerNorm over the FEATURE axis of x [C][T] (per token t: mean / biased variance over c, eps inside the sqrt),
// affine gamma / beta [C].  One block per 64 consecutive tokens, 16 waves: lane = token (coalesced rows), wave w owns the
// features w, w+16, ... and keeps them IN REGISTERS (one pass over x; all loads of a thread are independent and in
// flight together - the first version walked the 768 features of a token with 4 waves, one dependent-latency load
// after the other: 115 us per call on 100 tokens, 12 % of a swap).  The wave partials meet in LDS and are added in wave
// order (deterministic).  C <= 16 * kLnMaxV; wider inputs take the re-reading form below.
constexpr int kLnWaves = 16, kLnMaxV = 64;
__global__ __launch_bounds__(64 * kLnWaves) void channel_layernorm(float *__restrict__ out, const float *__restrict__ x,
                                                                   const float *__restrict__ gamma,
                                                                   const float *__restrict__ beta, int C, long long T,
                                                                   float eps) {
  HF_DYN_LDS;
  float *red = reinterpret_cast<float *>(hf_dyn_lds);  // [2][16 waves][64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long t = (long long)blockIdx.x * 64 + lane;
  const bool ok = t < T;
  const bool in_regs = C <= kLnWaves * kLnMaxV;
  float v[kLnMaxV];
  float s = 0.0f;
  if (in_regs) {
#pragma unroll
    for (int i = 0; i < kLnMaxV; ++i) {
      const int c = wave + i * kLnWaves;
      v[i] = (ok && c < C) ? x[(long long)c * T + t] : 0.0f;
      s += v[i];
    }
  } else {
    for (int c = wave; c < C; c += kLnWaves) s += ok ? x[(long long)c * T + t] : 0.0f;
  }
  red[wave * 64 + lane] = s;
  __syncthreads();
  float tot = 0.0f;
#pragma unroll
  for (int w = 0; w < kLnWaves; ++w) tot += red[w * 64 + lane];
  const float mean = tot / (float)C;
  float q = 0.0f;
  if (in_regs) {
#pragma unroll
    for (int i = 0; i < kLnMaxV; ++i) {
      const float d = (wave + i * kLnWaves < C) ? v[i] - mean : 0.0f;
      q = fmaf(d, d, q);
    }
  } else {
    for (int c = wave; c < C; c += kLnWaves) {
      const float d = ok ? x[(long long)c * T + t] - mean : 0.0f;
      q = fmaf(d, d, q);
    }
  }
  red[(kLnWaves + wave) * 64 + lane] = q;
  __syncthreads();
  float qt = 0.0f;
#pragma unroll
  for (int w = 0; w < kLnWaves; ++w) qt += red[(kLnWaves + w) * 64 + lane];
  const float inv = rsqrtf(qt / (float)C + eps);
  if (!ok) return;
  if (in_regs) {
#pragma unroll
    for (int i = 0; i < kLnMaxV; ++i) {
      const int c = wave + i * kLnWaves;
      if (c < C) {
        const float y = (v[i] - mean) * inv;
        out[(long long)c * T + t] = gamma ? fmaf(y, gamma[c], beta ? beta[c] : 0.0f) : y;
      }
    }
  } else {
    for (int c = wave; c < C; c += kLnWaves) {
      const float y = (x[(long long)c * T + t] - mean) * inv;
      out[(long long)c * T + t] = gamma ? fmaf(y, gamma[c], beta ? beta[c] : 0.0f) : y;
    }
  }
}

extern "C" int hf_channel_layernorm_f32(float *out, const float *x, const float *gamma, const float *beta, int channels,
                                        long long tokens, float eps, void *stream) {
  if (!out || !x || channels <= 0 || tokens <= 0) return HF_E_INVALID;
  hipLaunchKernelGGL(channel_layernorm, dim3(hf_cdiv(tokens, 64)), dim3(64 * kLnWaves), 2 * kLnWaves * 64 * sizeof(float),
                     (hipStream_t)stream, out, x, gamma, beta, channels, tokens, eps);
  return hf_launch_status();
}

// Multi-head self-attention of short sequences (nn.MultiheadAttention of ResidualAttentionBlock, no mask):
// qkv [3*E][T] feature-major (rows 0..E-1 = q, E..2E-1 = k, 2E..3E-1 = v; T = images * seq tokens, image-major),
// out [E][T]:  out[h*D + d][b*seq + i] = sum_j softmax_j(q_i . k_j / sqrt(D)) v_j[d].
// One block per (head, image); q, k, v of the head (feature-major, as they arrive: [D][sp]) and the TRANSPOSED score
// matrix sT[j][i] live in LDS - every inner-loop access is then either consecutive across the lanes or a broadcast.
// seq <= 64, D = 64: (3*64 + seq) * sp * 4 bytes <= 64 KiB.
constexpr int kMhaMaxSeq = 64, kMhaDim = 64;
__global__ __launch_bounds__(256) void mha_small(float *__restrict__ out, const float *__restrict__ qkv, int E, long long T,
                                                 int seq, int sp, float scale) {
  HF_DYN_LDS;
  float *q = reinterpret_cast<float *>(hf_dyn_lds);       // [D][sp]
  float *k = q + kMhaDim * sp;
  float *v = k + kMhaDim * sp;
  float *s = v + kMhaDim * sp;                             // [seq (j)][sp (i)]
  const int head = blockIdx.x, img = blockIdx.y;
  const long long t0 = (long long)img * seq;
  for (int i = threadIdx.x; i < kMhaDim * seq; i += 256) {  // coalesced over tokens
    const int d = i / seq, j = i - d * seq;
    const long long row = (long long)head * kMhaDim + d;
    q[d * sp + j] = qkv[row * T + t0 + j] * scale;
    k[d * sp + j] = qkv[(row + E) * T + t0 + j];
    v[d * sp + j] = qkv[(row + 2 * E) * T + t0 + j];
  }
  __syncthreads();
  for (int ij = threadIdx.x; ij < seq * seq; ij += 256) {
    const int i = ij / seq, j = ij - i * seq;
    float acc = 0.0f;
#pragma unroll 16
    for (int d = 0; d < kMhaDim; ++d) acc = fmaf(q[d * sp + i], k[d * sp + j], acc);
    s[j * sp + i] = acc;
  }
  __syncthreads();
  if (threadIdx.x < seq) {  // softmax of row i (seq <= 64 values: one thread per row)
    const int i = threadIdx.x;
    float m = s[i];
    for (int j = 1; j < seq; ++j) m = fmaxf(m, s[j * sp + i]);
    float z = 0.0f;
    for (int j = 0; j < seq; ++j) {
      const float e = expf(s[j * sp + i] - m);
      s[j * sp + i] = e;
      z += e;
    }
    const float inv = 1.0f / z;
    for (int j = 0; j < seq; ++j) s[j * sp + i] *= inv;
  }
  __syncthreads();
  for (int id = threadIdx.x; id < kMhaDim * seq; id += 256) {  // coalesced stores over tokens
    const int d = id / seq, i = id - d * seq;
    float acc = 0.0f;
    for (int j = 0; j < seq; ++j) acc = fmaf(s[j * sp + i], v[d * sp + j], acc);
    out[((long long)head * kMhaDim + d) * T + t0 + i] = acc;
  }
}

extern "C" int hf_mha_small_f32(float *out, const float *qkv, int images, int seq, int heads, int head_dim, void *stream) {
  if (!out || !qkv || images <= 0 || images > 65535 || seq <= 0 || seq > kMhaMaxSeq || heads <= 0 || head_dim != kMhaDim)
    return HF_E_INVALID;
  const int sp = (seq + 3) & ~3;
  const size_t lds = (size_t)(3 * kMhaDim + seq) * sp * sizeof(float);
  hipLaunchKernelGGL(mha_small, dim3(heads, images), dim3(256), lds, (hipStream_t)stream, out, qkv, heads * head_dim,
                     (long long)images * seq, seq, sp, 1.0f / sqrtf((float)head_dim));
  return hf_launch_status();
}

// QuickGELU (clip/model.py: x * sigmoid(1.702 x)), n elements
__global__ __launch_bounds__(256) void quick_gelu(float *__restrict__ out, const float *__restrict__ x, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const float v = x[i];
    out[i] = v / (1.0f + expf(-1.702f * v));
  }
}

extern "C" int hf_quick_gelu_f32(float *out, const float *x, long long n, void *stream) {
  if (!out || !x || n <= 0) return HF_E_INVALID;
  hipLaunchKernelGGL(quick_gelu, dim3(hf_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, out, x, n);
  return hf_launch_status();
}
