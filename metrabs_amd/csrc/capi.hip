// Version / error-string entry points of libmetrabs_hip.so.
#include "common.h"

extern "C" int mtr_version(void) { return MTR_VERSION; }

extern "C" const char* mtr_strerror(int code) {
  switch (code) {
    case MTR_OK: return "ok";
    case MTR_E_NULL: return "a required pointer is NULL";
    case MTR_E_SHAPE: return "a dimension is non-positive or outside the supported range";
    case MTR_E_DTYPE: return "unsupported dtype/layout combination";
    case MTR_E_PARAM: return "inconsistent params struct";
    case MTR_E_WORKSPACE: return "workspace too small or misaligned";
    case MTR_E_ALIGN: return "pointer violates the documented alignment";
    default: break;
  }
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "unknown metrabs_hip error";
}
