// Error reporting + ABI version for libspcl_hip.so (host only).
#include <stdarg.h>
#include <stdio.h>
#include "../../include/spcl_hip.h"

namespace spcl {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace spcl

extern "C" int spcl_abi_version(void) { return 1; }
extern "C" const char* spcl_last_error(void) { return spcl::g_err; }
