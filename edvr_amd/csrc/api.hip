// api.hip - error channel, version and device probe of libedvr_amd.so.
#include <cstring>

#include "common.h"

namespace edvr {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace edvr

extern "C" {
// A library built by scripts/build_variant.sh (kernel A/B experiments; the ablation switches of the kernel sources compute wrong
// results on purpose) carries its variant name in the version string: tests/conftest.py refuses to run the GPU suite on such a
// library, and a bench line records the string.
#ifdef EDVR_VARIANT
#define EDVR_STR2(x) #x
#define EDVR_STR(x) EDVR_STR2(x)
const char *edvr_version(void) { return "edvr_amd 0.4.0 (gfx950) variant:" EDVR_STR(EDVR_VARIANT); }
#else
const char *edvr_version(void) { return "edvr_amd 0.4.0 (gfx950)"; }
#endif
const char *edvr_last_error(void) { return edvr::g_err; }
int edvr_check_device(void) {
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    edvr::set_error("no HIP device");
    return EDVR_ERR_UNSUPPORTED;
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    edvr::set_error("device %s is not gfx950", prop.gcnArchName);
    return EDVR_ERR_UNSUPPORTED;
  }
  return EDVR_OK;
}
}
