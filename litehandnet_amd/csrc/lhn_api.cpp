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

// LHN_DETERMINISTIC=1 (read once): bit-reproducible results at a fraction of the speed.  Every accumulation that is shared
// between workgroups goes through REPLICAS indexed by blockIdx (32 for BatchNorm sums, 16 for weight gradients, 16 for the
// loss partials) that are folded in a fixed order afterwards; with at most 16 workgroups per launch each replica has ONE
// writer, so no sum depends on arrival order.  lhn_num_cus() is what every persistent grid is sized from: reporting 2 CUs
// (x at most 8 workgroups per CU) is the whole switch for those kernels; the per-sample attention-MLP backward kernels and
// the gate reduction, which add straight into one destination, run as a single ordered workgroup / one workgroup per image.
bool lhn_deterministic_mode() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("LHN_DETERMINISTIC");
    v = (e && e[0] == '1') ? 1 : 0;
  }
  return v == 1;
}

int lhn_num_cus() {
  if (lhn_deterministic_mode()) return 2;
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
int lhn_deterministic(void) { return lhn_deterministic_mode() ? 1 : 0; }
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
