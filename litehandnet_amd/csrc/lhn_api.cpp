// Error channel + version for the C ABI.
#include <stdarg.h>
#include "lhn_common.h"

static thread_local char g_err[512] = "";

void lhn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int lhn_num_cus() {
  static std::once_flag once[LHN_MAX_DEVICES];
  static int cus[LHN_MAX_DEVICES];
  const int slot = lhn_device_slot();
  std::call_once(once[slot], [&] {
    hipDeviceProp_t p;
    int d = 0, n = 0;
    if (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&p, d) == hipSuccess) n = p.multiProcessorCount;
    cus[slot] = n > 0 ? n : 256;
  });
  return cus[slot];
}

extern "C" {
int lhn_version(void) { return LHN_VERSION; }
const char* lhn_last_error(void) { return g_err; }
int lhn_device_ok(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    lhn_set_error("lhn_device_ok: no HIP device visible");
    return 1;
  }
  hipDeviceProp_t p;
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || hipGetDeviceProperties(&p, d) != hipSuccess) {
    lhn_set_error("lhn_device_ok: cannot query device");
    return 1;
  }
  if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
    lhn_set_error("lhn_device_ok: built for gfx950, device is %s", p.gcnArchName);
    return 1;
  }
  return 0;
}
}
