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

extern "C" int hf_abi_version(void) { return 10; }

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
