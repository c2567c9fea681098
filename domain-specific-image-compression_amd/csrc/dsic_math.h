// Deterministic special functions + the float32 table flow for the CDF tables of the
// entropy path (eval_selfcontained_entropy.py:14-15 gaussian_cdf, :17-23
// pmf_to_uint16_cdf, :41-47 z PMF, :57-58 StudentT.cdf).
//
// Bit-exactness between this GPU code and a CPU implementation cannot rest on
// libm / ocml (their exp, log, lgamma, erf differ in the last ulps, which the
// uint16 truncation of the tables, :22, can turn into a different byte
// stream).  Every routine here therefore uses only IEEE-754 +, -, *, / and
// comparisons in a fixed order (the library is built with -ffp-contract=off),
// so any implementation of the same sequence gives identical bits.  The
// algorithms are frozen in DESIGN.md ("Entropy path"):
//   exp    : k = round(x/ln2), Cody-Waite reduction, Taylor degree 14, scale by 2^k
//   log    : x = m 2^e, m in (sqrt(1/2), sqrt2], 2 atanh((m-1)/(m+1)) series, 12 terms
//   lgamma : recurrence up to x >= 12, Stirling series with 7 Bernoulli terms
//   Phi    : erfc by the e^{-u^2}-weighted erf series (u < 1.5) or the Laplace
//            continued fraction, fixed depth 120 (u >= 1.5)
//   I_x(a,b): modified-Lentz continued fraction (DLMF 8.17.22), <= 400 steps,
//            stop at |delta-1| < 3e-16, symmetry switch at x >= (a+1)/(a+b+2)
//   Student-t CDF: 1/2 +- 1/2 (1 - I_{nu/(nu+t^2)}(nu/2, 1/2))
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define DM_HD __host__ __device__ inline
#else
#define DM_HD inline
#endif

namespace dsic {
namespace dm {

#define DM_LN2_HI 6.93147180369123816490e-01
#define DM_LN2_LO 1.90821492927058770002e-10
#define DM_INV_LN2 1.44269504088896338700e+00
#define DM_SQRT2 1.41421356237309514547e+00
#define DM_INV_SQRT_PI 5.64189583547756279280e-01
#define DM_HALF_LOG_2PI 9.18938533204672780563e-01

DM_HD double from_bits(uint64_t b) {
  union {
    uint64_t u;
    double d;
  } v;
  v.u = b;
  return v.d;
}
DM_HD uint64_t to_bits(double d) {
  union {
    uint64_t u;
    double d;
  } v;
  v.d = d;
  return v.u;
}

DM_HD double scale2(double x, int k) {
  while (k > 1000) {
    x *= 8.98846567431157953865e+307;  // 2^1023
    k -= 1023;
  }
  while (k < -1000) {
    x *= 2.22507385850720138309e-308;  // 2^-1022
    k += 1022;
  }
  return x * from_bits((uint64_t)(k + 1023) << 52);
}

DM_HD double exp(double x) {
  if (x != x) return x;
  if (x > 709.0) return from_bits(0x7FF0000000000000ULL);
  if (x < -745.0) return 0.0;
  const double t = x * DM_INV_LN2;
  const int k = (int)(t < 0 ? t - 0.5 : t + 0.5);
  const double r = (x - (double)k * DM_LN2_HI) - (double)k * DM_LN2_LO;
  double p = 1.0 / 87178291200.0;
  p = p * r + 1.0 / 6227020800.0;
  p = p * r + 1.0 / 479001600;
  p = p * r + 1.0 / 39916800;
  p = p * r + 1.0 / 3628800;
  p = p * r + 1.0 / 362880;
  p = p * r + 1.0 / 40320;
  p = p * r + 1.0 / 5040;
  p = p * r + 1.0 / 720;
  p = p * r + 1.0 / 120;
  p = p * r + 1.0 / 24;
  p = p * r + 1.0 / 6;
  p = p * r + 0.5;
  p = p * r + 1.0;
  p = p * r + 1.0;
  return scale2(p, k);
}

DM_HD double log(double x) {
  if (x != x || x < 0) return from_bits(0x7FF8000000000000ULL);
  if (x == 0) return from_bits(0xFFF0000000000000ULL);
  int e = 0;
  if (x < 2.22507385850720138309e-308) {
    x *= 4503599627370496.0;
    e = -52;
  }
  uint64_t bits = to_bits(x);
  e += (int)((bits >> 52) & 0x7FF) - 1023;
  double m = from_bits((bits & 0x000FFFFFFFFFFFFFULL) | 0x3FF0000000000000ULL);
  if (m > DM_SQRT2) {
    m = m * 0.5;
    e += 1;
  }
  const double f = m - 1.0;
  const double s = f / (2.0 + f);
  const double z = s * s;
  double p = 1.0 / 25;
  for (int n = 11; n >= 1; --n) p = p * z + 1.0 / (double)(2 * n + 1);
  p = p * z + 1.0;
  const double de = (double)e;
  return de * DM_LN2_HI + (2.0 * s * p + de * DM_LN2_LO);
}

DM_HD double lgamma(double x) {
  double acc = 1.0;
  while (x < 12.0) {
    acc = acc * x;
    x = x + 1.0;
  }
  const double xi = 1.0 / x, x2 = xi * xi;
  double ser = 1.0 / 156;
  ser = ser * x2 - 691.0 / 360360;
  ser = ser * x2 + 1.0 / 1188;
  ser = ser * x2 - 1.0 / 1680;
  ser = ser * x2 + 1.0 / 1260;
  ser = ser * x2 - 1.0 / 360;
  ser = ser * x2 + 1.0 / 12;
  const double st = (x - 0.5) * log(x) - x + DM_HALF_LOG_2PI + ser * xi;
  return st - log(acc);
}

DM_HD double erfc_pos(double u) {
  if (u < 1.5) {
    const double u2 = u * u;
    double term = u, sum = u;
    for (int n = 1; n < 200; ++n) {
      term = term * (2.0 * u2) / (double)(2 * n + 1);
      sum = sum + term;
      if (term < sum * 1e-17) break;
    }
    return 1.0 - 2.0 * DM_INV_SQRT_PI * exp(-u2) * sum;
  }
  double d = u;
  for (int n = 120; n >= 1; --n) d = u + (0.5 * (double)n) / d;
  return exp(-u * u) * DM_INV_SQRT_PI / d;
}

DM_HD double normal_cdf(double x) {
  const double u = x / DM_SQRT2;
  if (u >= 0) return 1.0 - 0.5 * erfc_pos(u);
  return 0.5 * erfc_pos(-u);
}

DM_HD double betacf(double a, double b, double x) {
  const double TINY = 1e-300;
  const double qab = a + b, qap = a + 1.0, qam = a - 1.0;
  double c = 1.0, d = 1.0 - qab * x / qap;
  if (d < TINY && d > -TINY) d = TINY;
  d = 1.0 / d;
  double h = d;
  for (int m = 1; m <= 400; ++m) {
    const double dm_ = (double)m, m2 = 2.0 * dm_;
    double aa = dm_ * (b - dm_) * x / ((qam + m2) * (a + m2));
    d = 1.0 + aa * d;
    if (d < TINY && d > -TINY) d = TINY;
    c = 1.0 + aa / c;
    if (c < TINY && c > -TINY) c = TINY;
    d = 1.0 / d;
    h = h * d * c;
    aa = -(a + dm_) * (qab + dm_) * x / ((a + m2) * (qap + m2));
    d = 1.0 + aa * d;
    if (d < TINY && d > -TINY) d = TINY;
    c = 1.0 + aa / c;
    if (c < TINY && c > -TINY) c = TINY;
    d = 1.0 / d;
    const double del = d * c;
    h = h * del;
    double dev = del - 1.0;
    if (dev < 0) dev = -dev;
    if (dev < 3e-16) break;
  }
  return h;
}

DM_HD double betainc(double a, double b, double x, double xc) {
  if (x <= 0.0) return 0.0;
  if (xc <= 0.0) return 1.0;
  const double lbeta = lgamma(a + b) - lgamma(a) - lgamma(b);
  const double front = exp(lbeta + a * log(x) + b * log(xc));
  if (x < (a + 1.0) / (a + b + 2.0)) return front * betacf(a, b, x) / a;
  return 1.0 - front * betacf(b, a, xc) / b;
}

DM_HD double student_t_cdf(double t, double nu) {
  const double t2 = t * t;
  const double den = nu + t2;
  const double x = nu / den, xc = t2 / den;
  const double tail = 0.5 * betainc(0.5 * nu, 0.5, x, xc);
  return t > 0 ? 1.0 - tail : tail;
}

// ---- float32 table flow (eval_selfcontained_entropy.py:14-23, :41-47) ----------------------
// The reference evaluates the tables with float32 CPU-torch tensors; the steps below follow
// its operation order one by one (pinned by tests/golden/entropy_ref.npz, generated from the
// reference's own lines).  All float32 operations here are single IEEE operations
// (-ffp-contract=off, correctly rounded division), so host and device agree bit for bit.

// erf(u), u >= 0, float64: series below 1.5, 1 - Laplace continued fraction above.
DM_HD double erf_pos(double u) {
  if (u < 1.5) {
    const double u2 = u * u;
    double term = u, sum = u;
    for (int n = 1; n < 200; ++n) {
      term = term * (2.0 * u2) / (double)(2 * n + 1);
      sum = sum + term;
      if (term < sum * 1e-17) break;
    }
    return 2.0 * DM_INV_SQRT_PI * exp(-u2) * sum;
  }
  if (u > 6.0) return 1.0;
  return 1.0 - erfc_pos(u);
}

// float32 erf = the float64 value rounded once (<= 0.5 ulp).  torch's CPU erf (Intel MKL VML
// vsErf, closed source, < 1 ulp) differs from it in the last bit for ~1.5 % of arguments; that
// is the only step of the z-table flow that is not the reference's bits.
DM_HD float erff_rn(float a) {
  const double u = (double)a;
  if (u != u) return a;
  return u < 0 ? -(float)erf_pos(-u) : (float)erf_pos(u);
}

// :15 in float32: 0.5 * (1 + erf(x / float(sqrt 2)))
DM_HD float gaussian_cdf_f32(float x) {
  const float a = x / 1.41421354f;
  const float e = erff_rn(a);
  const float s = 1.0f + e;
  return 0.5f * s;
}

// torch.sum(dim=0) of float32 on the CPU, per output element (aten cascade sum): runs of 16
// added sequentially, run sums collected on level 1, level 1 flushed to level 2 at index
// multiples of 256, level 2 to level 3 at multiples of 4096; remainder, then levels low to high.
DM_HD float sum_f32_torch(const float* p, int n) {
  float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
  int i = 0;
  while (i + 16 <= n) {
    for (int j = 0; j < 16; ++j, ++i) a0 = a0 + p[i];
    a1 = a1 + a0;
    a0 = 0.0f;
    if ((i & 0xF0) == 0) {
      a2 = a2 + a1;
      a1 = 0.0f;
      if ((i & 0xF00) == 0) {
        a3 = a3 + a2;
        a2 = 0.0f;
      }
    }
  }
  for (; i < n; ++i) a0 = a0 + p[i];
  a0 = a0 + a1;
  a0 = a0 + a2;
  a0 = a0 + a3;
  return a0;
}

// Boundary CDF values F[0..L] (float32) -> coder table c[0..L-1] (uint16; c[L] = 65536 is
// implicit).  pmf clamp 1e-12 (:45), float32 sum + divide (:46), pmf_to_uint16_cdf (:17-23:
// cumsum with a float64 accumulator whose prefixes are rounded to float32, last >= 1,
// * 65535.0f, clamp, truncate), then the spreading c[k] = floor(u16[k] (65536-L) / 65535) + k
// that keeps every symbol's interval non-empty.  `pmf` holds L floats; raw (optional, may be
// null) receives the pre-spreading u16[0..L].
DM_HD void finish_table(const float* F, int L, uint16_t* out, float* pmf, uint16_t* raw) {
  for (int k = 0; k < L; ++k) {
    float p = F[k + 1] - F[k];
    if (p < 1e-12f) p = 1e-12f;
    pmf[k] = p;
  }
  const float total = sum_f32_torch(pmf, L);
  double cum = 0.0;
  uint32_t u16 = 0;
  for (int k = 0; k < L; ++k) {
    out[k] = (uint16_t)((uint32_t)(((uint64_t)u16 * (uint64_t)(65536 - L)) / 65535u) + (uint32_t)k);
    if (raw) raw[k] = (uint16_t)u16;
    const float q = pmf[k] / total;
    cum = cum + (double)q;
    float c = (float)cum;
    if (k == L - 1 && c < 1.0f) c = 1.0f;
    float sc = c * 65535.0f;
    if (sc < 0.0f) sc = 0.0f;
    if (sc > 65535.0f) sc = 65535.0f;
    u16 = (uint32_t)sc;
  }
  if (raw) raw[L] = (uint16_t)u16;
}

// Boundary k of a support starting at smin: float(smin + k) - 0.5f (= lower[k] = upper[k-1], :43).
DM_HD float boundary_f32(int smin, int k) { return (float)(smin + k) - 0.5f; }
DM_HD float table_cdf_gauss(int smin, int k, float sigma) {
  return gaussian_cdf_f32(boundary_f32(smin, k) / sigma);
}
// :57-58 (not executable in the reference): float64 t CDF of b/sigma, rounded once to float32.
DM_HD float table_cdf_student(int smin, int k, float sigma, float nu) {
  return (float)student_t_cdf((double)boundary_f32(smin, k) / (double)sigma, (double)nu);
}

}  // namespace dm
}  // namespace dsic
