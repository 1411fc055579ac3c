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
}  // namespace ms

extern "C" const char* ms_last_error(void) { return ms::g_err; }
extern "C" int ms_abi_version(void) { return MS_ABI_VERSION; }
