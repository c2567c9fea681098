// Thread-local last-error text for the C ABI.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include <atomic>

#include "common.h"

namespace dsic {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// The arithmetic variant of the contractions, one switch for the whole library (conv_first.hip, convT_image.hip;
// the host side picks the Winograd kernel family from the same value): 1 = operands split into two bf16 planes
// (default), 0 = fp32-input MFMAs.  The environment variable DSIC_WINO_BF16 only sets the initial value.
static std::atomic<int> g_split_bf16{-1};
int split_bf16() {
  int v = g_split_bf16.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* e = getenv("DSIC_WINO_BF16");
    v = (e && e[0] == '0' && e[1] == 0) ? 0 : 1;
    g_split_bf16.store(v, std::memory_order_relaxed);
  }
  return v;
}
}  // namespace dsic

extern "C" const char* dsic_last_error(void) { return dsic::g_err; }
extern "C" int dsic_abi_version(void) { return 4; }
extern "C" int dsic_split_bf16(void) { return dsic::split_bf16(); }
extern "C" int dsic_set_split_bf16(int on) {
  dsic::g_split_bf16.store(on ? 1 : 0, std::memory_order_relaxed);
  return DSIC_OK;
}
