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

extern "C" int hf_abi_version(void) { return 6; }
