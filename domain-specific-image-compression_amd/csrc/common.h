// Shared helpers for the libdsic_hip.so translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "dsic_hip.h"

namespace dsic {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return DSIC_EHIP;
  }
  return DSIC_OK;
}

#define DSIC_REQUIRE(cond, ...)          \
  do {                                   \
    if (!(cond)) {                       \
      ::dsic::set_error(__VA_ARGS__);    \
      return DSIC_EINVAL;                \
    }                                    \
  } while (0)

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

}  // namespace dsic
