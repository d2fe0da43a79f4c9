// api.hip - ABI bookkeeping entry points.
#include "hf_common.h"

extern "C" const char *hf_strerror(int code) {
  switch (code) {
    case HF_OK: return "ok";
    case HF_E_INVALID: return "invalid argument";
    case HF_E_LAUNCH: return "kernel launch failed";
    case HF_E_WORKSPACE: return "workspace too small";
    default: return "unknown error";
  }
}

extern "C" int hf_abi_version(void) { return 13; }

// A one-thread kernel with a name of its own: bench.py / tools launch it around the region a profile is about, so that
// rocprofv3's per-dispatch tables (kernel trace, counter collection) can be cut to that region - warm-up and plan-time
// kernels (weight splits, conv_prepare) excluded - by tools/summarize_prof.py.  `id` ends up in the grid size (id + 1 blocks).
__global__ void hf_profile_marker_kernel() {}

extern "C" int hf_profile_marker(int id, void *stream) {
  if (id < 0 || id > 1023) return HF_E_INVALID;
  hipLaunchKernelGGL(hf_profile_marker_kernel, dim3(id + 1), dim3(64), 0, static_cast<hipStream_t>(stream));
  return hipGetLastError() == hipSuccess ? HF_OK : HF_E_LAUNCH;
}

// per-translation-unit counters of the kernels that split fp32 into fp16 (hi, lo) pairs
extern "C" unsigned long long hf_f16_overflow_count_convh(int reset);
extern "C" unsigned long long hf_f16_overflow_count_blur(int reset);
extern "C" unsigned long long hf_f16_overflow_count_enc(int reset);
extern "C" unsigned long long hf_f16_overflow_count_gemm(int reset);
extern "C" unsigned long long hf_f16_overflow_count_stem(int reset);

extern "C" long long hf_f16_overflow_count(int reset) {
  return (long long)(hf_f16_overflow_count_convh(reset) + hf_f16_overflow_count_blur(reset) + hf_f16_overflow_count_enc(reset) +
                     hf_f16_overflow_count_gemm(reset) + hf_f16_overflow_count_stem(reset));
}
