/* ORACLE (test infrastructure, not product code): CPU restatement of the
 * per-patch entropy-coding path sketched in
 * code/modelv2/eval_selfcontained_entropy.py.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may build, load or call this file.
 *
 * Parity status: UNPINNED against reference bytes.  The reference script
 * cannot run on any torch (torch.distributions.StudentT.cdf raises
 * NotImplementedError at :57-58,110-111) and calls torchac 0.9.3
 * (Requirements.txt:237; third-party, not vendored, not installed) with
 * arguments that package does not accept.  No reference test or
 * fixture pins a single byte.  What is restated is the script's INTENT plus the
 * published torchac/L3C range coder, with the interpretation choices frozen in
 * DESIGN.md ("Entropy path"):
 *   (i)   support [min-tail, max+tail], L = max-min+2*tail+1          (:39-42,52-55)
 *   (ii)  z-PMF  = Phi((s+.5)/sigma_z) - Phi((s-.5)/sigma_z)           (:43-44)
 *         y-PMF  = StudentT(nu,0,sigma).cdf differences                (:57-58)
 *         both clamped at 1e-12 and renormalised                      (:45-46,59-60)
 *         evaluated in float64 with the routines below (+,-,*,/ only)
 *   (iii) pmf_to_uint16_cdf: cumsum, leading 0, last >= 1, *65535, clamp,
 *         truncate to uint16                                          (:17-23)
 *   (iv)  torchac-style spreading so every symbol keeps a non-empty interval:
 *         c[k] = floor(u16[k]*(65536-L)/65535) + k, c[L] = 65536 (implicit)
 *   (v)   symbols = value - min, scanned C, then H, then W            (:48,62)
 *   (vi)  coder: 32-bit low/high, 16-bit precision, MSB-first bits,
 *         pending-bit (E3) carry handling, final flush of pending+1 bits and
 *         zero padding to a byte (torchac_backend encode/decode).
 * An independent check of the math lives in tests/test_oracle_entropy.py
 * (scipy.special.ndtr / stdtr, and a pure-Python big-integer coder).
 *
 * Build: oracle/build_oracle.py (gcc -O2 -ffp-contract=off -shared).
 */
#include <stdint.h>
#include <string.h>

/* ------------------------------------------------------------------ math -- */
/* float64, operations restricted to + - * / and comparisons so that a second
 * implementation of the same sequence (the HIP kernel) is bit-identical. */

static const double LN2_HI = 6.93147180369123816490e-01;
static const double LN2_LO = 1.90821492927058770002e-10;
static const double INV_LN2 = 1.44269504088896338700e+00;
static const double SQRT2 = 1.41421356237309514547e+00;
static const double INV_SQRT_PI = 5.64189583547756279280e-01;
static const double HALF_LOG_2PI = 9.18938533204672780563e-01;

static double ora_ldexp(double x, int k) {
  /* x * 2^k by exponent-field arithmetic in up to three safe steps */
  while (k > 1000) { x *= 8.98846567431157953865e+307; /* 2^1023 */ k -= 1023; }
  while (k < -1000) { x *= 2.22507385850720138309e-308; /* 2^-1022 */ k += 1022; }
  uint64_t bits = (uint64_t)(k + 1023) << 52;
  double f;
  memcpy(&f, &bits, 8);
  return x * f;
}

double ora_exp(double x) {
  if (x != x) return x;
  if (x > 709.0) return 1.0 / 0.0;
  if (x < -745.0) return 0.0;
  double t = x * INV_LN2;
  int k = (int)(t < 0 ? t - 0.5 : t + 0.5);
  double r = (x - (double)k * LN2_HI) - (double)k * LN2_LO;
  /* Taylor to r^14, |r| <= 0.35 */
  static const double c[15] = {1.0, 1.0, 0.5, 1.0 / 6, 1.0 / 24, 1.0 / 120, 1.0 / 720, 1.0 / 5040,
                               1.0 / 40320, 1.0 / 362880, 1.0 / 3628800, 1.0 / 39916800,
                               1.0 / 479001600, 1.0 / 6227020800.0, 1.0 / 87178291200.0};
  double p = c[14];
  for (int i = 13; i >= 0; --i) p = p * r + c[i];
  return ora_ldexp(p, k);
}

double ora_log(double x) {
  if (x != x || x < 0) return 0.0 / 0.0;
  if (x == 0) return -1.0 / 0.0;
  int e = 0;
  if (x < 2.22507385850720138309e-308) { x *= 4503599627370496.0; e = -52; }
  uint64_t bits;
  memcpy(&bits, &x, 8);
  e += (int)((bits >> 52) & 0x7FF) - 1023;
  bits = (bits & 0x000FFFFFFFFFFFFFULL) | 0x3FF0000000000000ULL;
  double m;
  memcpy(&m, &bits, 8);
  if (m > SQRT2) { m = m * 0.5; e += 1; }
  double f = m - 1.0;
  double s = f / (2.0 + f);
  double z = s * s;
  /* 2*atanh(s) = 2s(1 + z/3 + z^2/5 + ...), z <= 0.0295: 12 terms */
  double p = 1.0 / 25;
  for (int n = 11; n >= 1; --n) p = p * z + 1.0 / (double)(2 * n + 1);
  p = p * z + 1.0;
  double de = (double)e;
  return de * LN2_HI + (2.0 * s * p + de * LN2_LO);
}

double ora_lgamma(double x) {
  /* x > 0.  Shift to x >= 12, Stirling series with 7 correction terms. */
  double acc = 1.0;
  while (x < 12.0) { acc = acc * x; x = x + 1.0; }
  double xi = 1.0 / x, x2 = xi * xi;
  double ser = 1.0 / 156;
  ser = ser * x2 - 691.0 / 360360;
  ser = ser * x2 + 1.0 / 1188;
  ser = ser * x2 - 1.0 / 1680;
  ser = ser * x2 + 1.0 / 1260;
  ser = ser * x2 - 1.0 / 360;
  ser = ser * x2 + 1.0 / 12;
  double st = (x - 0.5) * ora_log(x) - x + HALF_LOG_2PI + ser * xi;
  return st - ora_log(acc);
}

static double ora_erfc_pos(double u) {
  /* erfc(u), u >= 0 */
  if (u < 1.5) {
    /* erf(u) = 2/sqrt(pi) e^{-u^2} sum_n 2^n u^(2n+1) / (1*3*...*(2n+1)) */
    double u2 = u * u, term = u, sum = u;
    for (int n = 1; n < 200; ++n) {
      term = term * (2.0 * u2) / (double)(2 * n + 1);
      sum = sum + term;
      if (term < sum * 1e-17) break;
    }
    return 1.0 - 2.0 * INV_SQRT_PI * ora_exp(-u2) * sum;
  }
  /* continued fraction: erfc(u) = e^{-u^2}/sqrt(pi) * 1/(u + (1/2)/(u + 1/(u + (3/2)/(u + ...)))) */
  double d = u;
  for (int n = 120; n >= 1; --n) d = u + (0.5 * (double)n) / d;
  return ora_exp(-u * u) * INV_SQRT_PI / d;
}

double ora_normal_cdf(double x) {
  /* Phi(x) = 1/2 (1 + erf(x/sqrt2))  (eval_selfcontained_entropy.py:14-15) */
  double u = x / SQRT2;
  if (u >= 0) return 1.0 - 0.5 * ora_erfc_pos(u);
  return 0.5 * ora_erfc_pos(-u);
}

static double ora_betacf(double a, double b, double x) {
  /* modified Lentz evaluation of the continued fraction of I_x(a,b) (DLMF 8.17.22) */
  const double TINY = 1e-300;
  double qab = a + b, qap = a + 1.0, qam = a - 1.0;
  double c = 1.0, d = 1.0 - qab * x / qap;
  if (d < TINY && d > -TINY) d = TINY;
  d = 1.0 / d;
  double h = d;
  for (int m = 1; m <= 400; ++m) {
    double dm = (double)m, m2 = 2.0 * dm;
    double aa = dm * (b - dm) * x / ((qam + m2) * (a + m2));
    d = 1.0 + aa * d;
    if (d < TINY && d > -TINY) d = TINY;
    c = 1.0 + aa / c;
    if (c < TINY && c > -TINY) c = TINY;
    d = 1.0 / d;
    h = h * d * c;
    aa = -(a + dm) * (qab + dm) * x / ((a + m2) * (qap + m2));
    d = 1.0 + aa * d;
    if (d < TINY && d > -TINY) d = TINY;
    c = 1.0 + aa / c;
    if (c < TINY && c > -TINY) c = TINY;
    d = 1.0 / d;
    double del = d * c;
    h = h * del;
    double dev = del - 1.0;
    if (dev < 0) dev = -dev;
    if (dev < 3e-16) break;
  }
  return h;
}

static double ora_betainc(double a, double b, double x, double xc) {
  /* regularised incomplete beta I_x(a,b); xc = 1-x supplied by the caller */
  if (x <= 0.0) return 0.0;
  if (xc <= 0.0) return 1.0;
  double lbeta = ora_lgamma(a + b) - ora_lgamma(a) - ora_lgamma(b);
  double front = ora_exp(lbeta + a * ora_log(x) + b * ora_log(xc));
  if (x < (a + 1.0) / (a + b + 2.0)) return front * ora_betacf(a, b, x) / a;
  return 1.0 - front * ora_betacf(b, a, xc) / b;
}

double ora_student_t_cdf(double t, double nu) {
  /* F(t) = 1/2 + 1/2 sign(t) (1 - I_{nu/(nu+t^2)}(nu/2, 1/2)) */
  double t2 = t * t;
  double den = nu + t2;
  double x = nu / den, xc = t2 / den;
  double tail = 0.5 * ora_betainc(0.5 * nu, 0.5, x, xc);
  return t > 0 ? 1.0 - tail : tail;
}

/* ---------------------------------------------------------------- tables -- */

/* Shared tail: boundaries' CDF values F[0..L] -> coder table c[0..L-1]. */
static void ora_finish_table(const double* F, int L, uint16_t* out, double* work) {
  double total = 0.0;
  for (int k = 0; k < L; ++k) {
    double p = F[k + 1] - F[k];
    if (p < 1e-12) p = 1e-12;
    work[k] = p;
    total = total + p;
  }
  double cum = 0.0;
  for (int k = 0; k <= L; ++k) {
    /* cdf_with_zero[k] */
    double v = cum;
    if (k == L && v < 1.0) v = 1.0;
    double sc = v * 65535.0;
    if (sc < 0.0) sc = 0.0;
    if (sc > 65535.0) sc = 65535.0;
    uint32_t u16 = (uint32_t)sc; /* truncation, :22 */
    uint32_t ck = (uint32_t)(((uint64_t)u16 * (uint64_t)(65536 - L)) / 65535u) + (uint32_t)k;
    if (k < L) {
      out[k] = (uint16_t)ck;
      cum = cum + work[k] / total;
    }
  }
}

/* tables: [C][L] uint16.  sigma: [C] float32 (= exp(log_sigma), no clamp: :32). */
void ora_tables_gauss(const float* sigma, int C, int smin, int L, uint16_t* tables, double* work) {
  double* F = work;          /* L+1 */
  double* w2 = work + L + 1; /* L */
  for (int c = 0; c < C; ++c) {
    double sg = (double)sigma[c];
    for (int k = 0; k <= L; ++k) F[k] = ora_normal_cdf(((double)(smin + k) - 0.5) / sg);
    ora_finish_table(F, L, tables + (size_t)c * L, w2);
  }
}

void ora_tables_student(const float* sigma, const float* nu, int C, int smin, int L, uint16_t* tables,
                        double* work) {
  double* F = work;
  double* w2 = work + L + 1;
  for (int c = 0; c < C; ++c) {
    double sg = (double)sigma[c], nv = (double)nu[c];
    for (int k = 0; k <= L; ++k) F[k] = ora_student_t_cdf((((double)(smin + k) - 0.5) - 0.0) / sg, nv);
    ora_finish_table(F, L, tables + (size_t)c * L, w2);
  }
}

/* ----------------------------------------------------------- range coder -- */

typedef struct {
  uint8_t* out;
  int64_t cap, n;
  uint32_t cache;
  int count;
  int overflow;
} ora_bits;

static void put_bit(ora_bits* b, int bit) {
  b->cache = (b->cache << 1) | (uint32_t)bit;
  b->count += 1;
  if (b->count == 8) {
    if (b->n < b->cap) b->out[b->n] = (uint8_t)b->cache; else b->overflow = 1;
    b->n += 1;
    b->cache = 0;
    b->count = 0;
  }
}

static void put_bit_and_pending(ora_bits* b, int bit, uint64_t* pending) {
  put_bit(b, bit);
  while (*pending > 0) { put_bit(b, !bit); *pending -= 1; }
}

/* sym[n] in [0,L), channel of symbol i = i / hw.  Returns bytes written (or -1 on overflow). */
int64_t ora_range_encode(const int32_t* sym, int64_t n, const uint16_t* tables, int L, int hw,
                         uint8_t* out, int64_t cap) {
  ora_bits b = {out, cap, 0, 0, 0, 0};
  uint32_t low = 0, high = 0xFFFFFFFFu;
  uint64_t pending = 0;
  for (int64_t i = 0; i < n; ++i) {
    const uint16_t* t = tables + (size_t)(i / hw) * L;
    int s = sym[i];
    uint64_t c_low = t[s];
    uint64_t c_high = (s == L - 1) ? 0x10000u : t[s + 1];
    uint64_t span = (uint64_t)high - (uint64_t)low + 1;
    high = (low - 1) + (uint32_t)((span * c_high) >> 16);
    low = low + (uint32_t)((span * c_low) >> 16);
    for (;;) {
      if (high < 0x80000000u) {
        put_bit_and_pending(&b, 0, &pending);
        low <<= 1; high <<= 1; high |= 1;
      } else if (low >= 0x80000000u) {
        put_bit_and_pending(&b, 1, &pending);
        low <<= 1; high <<= 1; high |= 1;
      } else if (low >= 0x40000000u && high < 0xC0000000u) {
        pending += 1;
        low <<= 1; low &= 0x7FFFFFFFu;
        high <<= 1; high |= 0x80000001u;
      } else {
        break;
      }
    }
  }
  pending += 1;
  put_bit_and_pending(&b, low < 0x40000000u ? 0 : 1, &pending);
  if (b.count > 0) while (b.count != 0) put_bit(&b, 0);
  return b.overflow ? -1 : b.n;
}

typedef struct {
  const uint8_t* in;
  int64_t n, pos;
  int bit;
} ora_in;

static uint32_t get_bit(ora_in* r) {
  if (r->pos >= r->n) return 0;
  uint32_t v = (r->in[r->pos] >> (7 - r->bit)) & 1u;
  r->bit += 1;
  if (r->bit == 8) { r->bit = 0; r->pos += 1; }
  return v;
}

void ora_range_decode(const uint8_t* in, int64_t nbytes, int64_t n, const uint16_t* tables, int L,
                      int hw, int32_t* sym) {
  ora_in r = {in, nbytes, 0, 0};
  uint32_t low = 0, high = 0xFFFFFFFFu, value = 0;
  for (int i = 0; i < 32; ++i) value = (value << 1) | get_bit(&r);
  for (int64_t i = 0; i < n; ++i) {
    const uint16_t* t = tables + (size_t)(i / hw) * L;
    uint64_t span = (uint64_t)high - (uint64_t)low + 1;
    uint32_t count = (uint32_t)(((((uint64_t)value - (uint64_t)low + 1) << 16) - 1) / span);
    /* largest s with c[s] <= count */
    int lo = 0, hi = L - 1;
    while (lo < hi) {
      int mid = (lo + hi + 1) >> 1;
      if ((uint32_t)t[mid] <= count) lo = mid; else hi = mid - 1;
    }
    int s = lo;
    sym[i] = s;
    if (i == n - 1) break;
    uint64_t c_low = t[s];
    uint64_t c_high = (s == L - 1) ? 0x10000u : t[s + 1];
    high = (low - 1) + (uint32_t)((span * c_high) >> 16);
    low = low + (uint32_t)((span * c_low) >> 16);
    for (;;) {
      if (low >= 0x80000000u || high < 0x80000000u) {
        low <<= 1; high <<= 1; high |= 1;
        value = (value << 1) | get_bit(&r);
      } else if (low >= 0x40000000u && high < 0xC0000000u) {
        low <<= 1; low &= 0x7FFFFFFFu;
        high <<= 1; high |= 0x80000001u;
        value -= 0x40000000u;
        value = (value << 1) | get_bit(&r);
      } else {
        break;
      }
    }
  }
}
