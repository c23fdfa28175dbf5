// capi.hip — library identification and error naming for the C ABI (include/lgu_corr.h).
#include "lgu_common.hpp"

extern "C" {

const char* lgu_version(void) { return "lgu_corr 0.6.0 gfx950"; }

const char* lgu_error_string(int code) {
  switch (code) {
    case LGU_OK: return "success";
    case LGU_E_BADARG: return "lgu: bad argument (null pointer, non-positive size or radius out of range)";
    case LGU_E_UNSUPPORTED: return "lgu: shape not served by the gfx950 kernels";
    default: return hipGetErrorString(static_cast<hipError_t>(code));
  }
}

}  // extern "C"
