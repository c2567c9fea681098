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


#if defined(__HIPCC__)
// GDN / IGDN of one value (layers.py:21-27): x / sqrt(beta + gamma x^2) or x * sqrt(...).
// DSIC_EXACT_GDN=1 reproduces torch's op sequence with IEEE div/sqrt (~25 VALU ops per
// element); the default uses v_rsq_f32 plus one Newton step (~9 ops, |rel. error| < 2^-22),
// far below the fp32 summation-order noise of the convolution that feeds it.
#ifndef DSIC_EXACT_GDN
#define DSIC_EXACT_GDN 0
#endif
__device__ __forceinline__ float gdn_apply(float v, float beta, float gamma, bool inverse) {
  const float s = __fadd_rn(beta, __fmul_rn(gamma, __fmul_rn(v, v)));
#if DSIC_EXACT_GDN
  const float d = __fsqrt_rn(s);
  return inverse ? __fmul_rn(v, d) : __fdiv_rn(v, d);
#else
  float r = __builtin_amdgcn_rsqf(s);
  r = r * (1.5f - 0.5f * s * r * r);  // Newton step on 1/sqrt(s)
  return inverse ? v * (s * r) : v * r;
#endif
}
#endif

}  // namespace dsic
