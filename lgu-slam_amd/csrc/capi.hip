// capi.hip — library identification, error naming and the debug-knob gate for the C ABI (include/lgu_corr.h).
#include <string.h>

#include "lgu_common.hpp"

#ifndef LGU_BUILD_FLAGS
#define LGU_BUILD_FLAGS ""
#endif

namespace lgu {

// Read once, when the library is loaded (dynamic initialisation at dlopen); never written again.
static const bool g_debug_knobs = [] {
  const char* s = getenv("LGU_DEBUG_KNOBS");
  return s != nullptr && strcmp(s, "1") == 0;
}();

bool debug_knobs() { return g_debug_knobs; }

}  // namespace lgu

extern "C" {

// "lgu_corr <version> gfx950", followed by the extra compiler flags in brackets when the library is not the default
// build (lgu-slam_amd/_build.py, LGU_EXTRA_HIPCC_FLAGS — experiments only): evidence records which build produced it.
const char* lgu_version(void) {
  return sizeof(LGU_BUILD_FLAGS) > 1 ? "lgu_corr 0.7.0 gfx950 [" LGU_BUILD_FLAGS "]" : "lgu_corr 0.7.0 gfx950";
}

int lgu_debug_knobs_enabled(void) { return lgu::g_debug_knobs ? 1 : 0; }

const char* lgu_error_string(int code) {
  switch (code) {
    case LGU_OK: return "success";
    case LGU_E_BADARG: return "lgu: bad argument (null pointer, non-positive size or radius out of range)";
    case LGU_E_UNSUPPORTED: return "lgu: shape not served by the gfx950 kernels";
    default: return hipGetErrorString(static_cast<hipError_t>(code));
  }
}

}  // extern "C"
