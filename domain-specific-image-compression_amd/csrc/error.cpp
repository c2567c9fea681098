// Thread-local last-error text for the C ABI.
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

namespace dsic {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace dsic

extern "C" const char* dsic_last_error(void) { return dsic::g_err; }
extern "C" int dsic_abi_version(void) { return 3; }
