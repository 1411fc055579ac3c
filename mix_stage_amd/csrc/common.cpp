// Error reporting + trivial entry points of libmixstage_hip.so.
#include "common.h"

namespace ms {
static thread_local char g_err[512] = "";

int set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return -1;
}
int* g_counters = nullptr;
int g_counters_n = 0;
}  // namespace ms

extern "C" const char* ms_last_error(void) { return ms::g_err; }
extern "C" int ms_abi_version(void) { return MS_ABI_VERSION; }

extern "C" int ms_set_counter_buffer(int32_t* zeroed_counters, int n) {
  ms::g_counters = zeroed_counters;
  ms::g_counters_n = zeroed_counters ? n : 0;
  return 0;
}

namespace ms {
int current_device_index() {
  int dev = -1;
  return hipGetDevice(&dev) == hipSuccess ? dev : -1;
}
int current_device_cus() {
  static int cus[64] = {0};
  const int dev = current_device_index();
  if (dev < 0 || dev > 63) return 0;
  if (!cus[dev]) {
    hipDeviceProp_t prop;
    cus[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : -1;
  }
  return cus[dev] > 0 ? cus[dev] : 0;
}
}  // namespace ms
