// Per-patch entropy coding on the GPU: support scan, CDF tables, range
// encoder / decoder.  Restates the intent of
// code/modelv2/eval_selfcontained_entropy.py:14-23 (gaussian_cdf,
// pmf_to_uint16_cdf), :36-62 (per-image support, PMFs, z-then-y strings) and
// :86-116 (decode), with the torchac 0.9.3 coder it calls (:48,62,96,116;
// third-party: 32-bit low/high, 16-bit precision, pending-bit carry
// resolution).  Interpretation choices are frozen in DESIGN.md "Entropy path".
//
// Parallel structure: tables are embarrassingly parallel (one wave per
// (image, channel) table: 64 lanes evaluate the boundary CDFs, lane 0 does the
// short sequential normalise/cumsum/quantise in the reference's float32 order).  The coder's interval
// update is a serial dependency chain per stream, so there is one workgroup
// (one wave) per (image, stream): all 64 lanes translate a chunk of symbols to
// (c_low, c_high) pairs in LDS, then lane 0 walks the chunk.  Matching leading
// bits and pending (E3) runs are shifted out in bulk with clz instead of bit
// by bit; the emitted bytes are identical to the bit-serial reference loop.
// Integer work only after the tables: results are bit-exact by construction.
#include "common.h"
#include "dsic_math.h"

namespace dsic {

// meta[b] = {ymin - tail, Ly, zmin - tail, Lz}: :39-41, :52-54 (values are
// integers already, so floor/ceil are the identity).
__global__ __launch_bounds__(256) void support_kernel(const float* __restrict__ y,
                                                      const float* __restrict__ z,
                                                      int* __restrict__ meta, int64_t ny, int64_t nz,
                                                      int tail) {
  __shared__ float red[2][4];
  __shared__ int bad_any;
  const int b = blockIdx.x;
  for (int which = 0; which < 2; ++which) {
    const float* p = which ? z + (size_t)b * nz : y + (size_t)b * ny;
    const int64_t n = which ? nz : ny;
    float mn = p[0], mx = p[0];
    bool bad = false;  // NaN, infinity or a magnitude no int support can hold
    for (int64_t i = threadIdx.x; i < n; i += 256) {
      const float v = p[i];
      bad |= !(fabsf(v) < 1.0e9f);
      mn = fminf(mn, v);
      mx = fmaxf(mx, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mn = fminf(mn, __shfl_down(mn, o, 64));
      mx = fmaxf(mx, __shfl_down(mx, o, 64));
    }
    if (threadIdx.x == 0) bad_any = 0;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
      red[0][threadIdx.x >> 6] = mn;
      red[1][threadIdx.x >> 6] = mx;
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(&bad_any, 1);
    __syncthreads();
    if (threadIdx.x == 0) {
      mn = fminf(fminf(red[0][0], red[0][1]), fminf(red[0][2], red[0][3]));
      mx = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
      if (bad_any) {
        // non-finite latents have no support: width 0 makes every consumer of `meta` raise err bit 1
        meta[4 * b + 2 * which] = 0;
        meta[4 * b + 2 * which + 1] = 0;
      } else {
        const int lo = (int)floorf(mn) - tail, hi = (int)ceilf(mx) + tail;
        meta[4 * b + 2 * which] = lo;
        meta[4 * b + 2 * which + 1] = hi - lo + 1;
      }
    }
    __syncthreads();
  }
}

// dm::finish_table with the 64 lanes of a wave: everything that is element-wise (pmf, clamp, divide, scaling,
// truncation, spreading) runs one entry per lane; the two order-sensitive reductions keep the serial order of the
// host function - the float32 sum (torch's cascade) is evaluated by every lane alike, the float64 running sum walks
// the entries through v_readlane.  Same operations on the same operands: the tables are bit-identical to
// dm::finish_table (tests/test_gpu_entropy.py compares them with the host entry point).  F[0..L] is overwritten
// with the rounded prefixes.
__device__ __forceinline__ void finish_table_wave(float* F, int L, uint16_t* out, float* pmf, int lane) {
  for (int k = lane; k < L; k += 64) {
    float p = F[k + 1] - F[k];
    if (p < 1e-12f) p = 1e-12f;
    pmf[k] = p;
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  const float total = dm::sum_f32_torch(pmf, L);
  double cum = 0.0;
  for (int base = 0; base < L; base += 64) {
    const int k = base + lane;
    const float q = k < L ? pmf[k] / total : 0.0f;
    const int n = L - base < 64 ? L - base : 64;
    float mine = 0.0f;
    for (int j = 0; j < n; ++j) {
      const float qj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, q), j));
      cum = cum + (double)qj;
      float c = (float)cum;
      if (base + j == L - 1 && c < 1.0f) c = 1.0f;
      if (j == lane) mine = c;
    }
    if (k < L) F[k] = mine;       // prefix that includes entry k
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  for (int k = lane; k < L; k += 64) {
    uint32_t u16 = 0;
    if (k > 0) {
      float sc = F[k - 1] * 65535.0f;
      if (sc < 0.0f) sc = 0.0f;
      if (sc > 65535.0f) sc = 65535.0f;
      u16 = (uint32_t)sc;
    }
    out[k] = (uint16_t)((u16 * (uint32_t)(65536 - L)) / 65535u + (uint32_t)k);   // < 2^32: one 32-bit division
  }
}

// One wave per table.  sigma/nu indexed [b*sb + c] (sb = 0 for the image-independent z prior).
template <bool STUDENT>
__global__ __launch_bounds__(256) void tables_kernel(const float* __restrict__ sigma,
                                                     const float* __restrict__ nu, int sb,
                                                     const int* __restrict__ meta, int meta_off,
                                                     uint16_t* __restrict__ tables, int C, int Lmax,
                                                     int ntables, int* __restrict__ err) {
  extern __shared__ float shm[];  // per wave: F[Lmax+1], pmf[Lmax]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tid = blockIdx.x * 4 + wave;
  if (tid >= ntables) return;
  const int b = tid / C, c = tid % C;
  const int smin = meta[4 * b + meta_off], L = meta[4 * b + meta_off + 1];
  if (L > Lmax || L < 1) {
    if (lane == 0) atomicOr(err, 1);
    return;
  }
  float* F = shm + (size_t)wave * (2 * Lmax + 1);
  float* pmf = F + Lmax + 1;
  const float sg = sigma[(size_t)b * sb + c];
  const float nv = STUDENT ? nu[(size_t)b * sb + c] : 0.0f;
  for (int k = lane; k <= L; k += 64)
    F[k] = STUDENT ? dm::table_cdf_student(smin, k, sg, nv) : dm::table_cdf_gauss(smin, k, sg);
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  finish_table_wave(F, L, tables + ((size_t)b * C + c) * Lmax, pmf, lane);
}

// Range encoder, one wave per (image, stream): stream ids 0..B-1 = y strings, B..2B-1 = z strings (which 0: z string,
// 1: y string); several streams (one per SIMD) share a workgroup.
//
// The coder is split along its only true dependency.  (1) The interval recurrence
// (low, high, pending) is a serial chain per stream: it runs branch-free on wave-uniform
// values (scalar unit), reading the (c_low, c_high-1) pair of symbol j with v_readlane from a
// register the 64 lanes filled in parallel (table gather, pipelined two groups ahead), and
// records per symbol only WHAT is emitted: the nb leading bits of low that became final and
// the pending count flushed behind the first of them (v_writelane).  (2) WHERE those bits go
// is a prefix sum: every 64 symbols the lanes scan their bit counts and OR their pieces into
// the zero-initialised output with atomics, in parallel.  Bytes are identical to the
// bit-serial reference loop (torchac): E1/E2 shifts = clz(low^high) at once, E3 run = leading
// ones of (low<<1)&~(high<<1).
__device__ __forceinline__ void put_bits(uint32_t* out32, int64_t cap_bits, int64_t off, uint32_t v,
                                         int len, int* overflow) {
  // append the low `len` (1..32) bits of v at bit offset `off`, MSB first
  if (off + len > cap_bits) {
    *overflow = 1;
    return;
  }
  const int64_t w = off >> 5;
  const int s = (int)(off & 31), space = 32 - s;
  if (len <= space) {
    atomicOr(out32 + w, __builtin_bswap32(v << (space - len)));
  } else {
    const int rest = len - space;
    atomicOr(out32 + w, __builtin_bswap32(v >> rest));
    atomicOr(out32 + w + 1, __builtin_bswap32(v << (32 - rest)));
  }
}

__device__ __forceinline__ void put_ones(uint32_t* out32, int64_t cap_bits, int64_t off, uint32_t count,
                                         int* overflow) {
  while (count > 0) {
    const int len = count > 32 ? 32 : (int)count;
    put_bits(out32, cap_bits, off, len == 32 ? 0xFFFFFFFFu : ((1u << len) - 1u), len, overflow);
    off += len;
    count -= len;
  }
}

// ---- fast serial step (full groups of 64 symbols whose c_high < 65536) ------------------------
// State (low, span) with span = high - low + 1 kept mod 2^32 (0 means 2^32).  The lanes hold
// c << 16, so floor(span * c / 2^16) is ONE s_mul_hi_u32; the step is unrolled with immediate
// lane selects (no M0 / wait-state padding), and only (low1, high1) — the interval BEFORE
// renormalisation — are recorded: which bits became final (E1/E2), how long the E3 run is and
// what is owed from earlier symbols is recomputed from them by the 64 lanes in parallel.
// ~22 scalar instructions per symbol instead of ~55.
#ifndef ENC_SELECT
#define ENC_SELECT 1   // 1: s_cselect for the span = 2^32 case (4 % faster than a branch per symbol)
#endif
template <int J>
__device__ __forceinline__ void wlane(uint32_t& rec, uint32_t v) {
  asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(rec) : "s"(v), "n"(J));
}

template <int J>
__device__ __forceinline__ uint32_t rlane(uint32_t v) {
  uint32_t r;
  asm volatile("v_readlane_b32 %0, %1, %2" : "=s"(r) : "v"(v), "n"(J));
  return r;
}

// (cl, ch): the bounds of symbol J, read one step earlier; the step first reads those of symbol J+1,
// so the VALU -> SGPR latency of v_readlane never sits on the serial chain.
template <int J>
__device__ __forceinline__ void enc_step(uint32_t& low, uint32_t& lowm1, uint32_t& span, uint32_t& cl,
                                         uint32_t& ch, uint32_t clo16, uint32_t chi16, uint32_t& rec_low,
                                         uint32_t& rec_high) {
  // The chain span -> span of the next symbol is ~9 dependent scalar operations (mul_hi, select,
  // sub/add, xor | orn2, flbit | shift, shift, flbit, add, shift); low - 1 and the record stores sit
  // beside it.
  const uint32_t cl_next = J < 63 ? rlane<(J < 63 ? J + 1 : 63)>(clo16) : 0u;
  const uint32_t ch_next = J < 63 ? rlane<(J < 63 ? J + 1 : 63)>(chi16) : 0u;
  uint32_t lo_add, hi_add;
#if ENC_SELECT
  {  // branch-free: span = 2^32 (kept as 0) selects (2^32 c) >> 16 = c << 16
    const uint32_t ml = __umulhi(span, cl), mh = __umulhi(span, ch);
    lo_add = span ? ml : cl;
    hi_add = span ? mh : ch;
  }
#else
  if (__builtin_expect(span == 0u, 0)) {  // span = 2^32: (2^32 c) >> 16
    lo_add = cl;
    hi_add = ch;
  } else {
    lo_add = __umulhi(span, cl);
    hi_add = __umulhi(span, ch);
  }
#endif
  const uint32_t low1 = low + lo_add;
  const uint32_t high1 = lowm1 + hi_add;         // low - 1 + floor(span c_high / 2^16)
  const uint32_t span1 = hi_add - lo_add;        // >= 2^14 - 1
  wlane<J>(rec_low, low1);
  wlane<J>(rec_high, high1);
  const int nb = __builtin_clz(low1 ^ high1);    // E1/E2
  const uint32_t q2 = (high1 | ~low1) << 1;      // 0 where (low, high) = (1, 0): E3 pairs below the split bit
  const int m = __builtin_clz(q2 << nb);         // != 0: an all-E3 tail would need span1 == 2
  const int sh = nb + m;
  span = span1 << sh;                            // exactly 2^32 -> 0
  low = (low1 << sh) & 0x7FFFFFFFu;
  lowm1 = low - 1u;
  cl = cl_next;
  ch = ch_next;
}

template <int J0>
__device__ __forceinline__ void enc_steps8(uint32_t& low, uint32_t& lowm1, uint32_t& span, uint32_t& cl,
                                           uint32_t& ch, uint32_t clo16, uint32_t chi16, uint32_t& rec_low,
                                           uint32_t& rec_high) {
  enc_step<J0 + 0>(low, lowm1, span, cl, ch, clo16, chi16, rec_low, rec_high);
  enc_step<J0 + 1>(low, lowm1, span, cl, ch, clo16, chi16, rec_low, rec_high);
  enc_step<J0 + 2>(low, lowm1, span, cl, ch, clo16, chi16, rec_low, rec_high);
  enc_step<J0 + 3>(low, lowm1, span, cl, ch, clo16, chi16, rec_low, rec_high);
  enc_step<J0 + 4>(low, lowm1, span, cl, ch, clo16, chi16, rec_low, rec_high);
  enc_step<J0 + 5>(low, lowm1, span, cl, ch, clo16, chi16, rec_low, rec_high);
  enc_step<J0 + 6>(low, lowm1, span, cl, ch, clo16, chi16, rec_low, rec_high);
  enc_step<J0 + 7>(low, lowm1, span, cl, ch, clo16, chi16, rec_low, rec_high);
}

__global__ __launch_bounds__(1024) void range_encode_kernel(
    const float* __restrict__ y, const float* __restrict__ z, const int* __restrict__ meta,
    const uint16_t* __restrict__ tab_y, const uint16_t* __restrict__ tab_z, int Lmax, int M, int HWy,
    int N, int HWz, uint8_t* __restrict__ out, int64_t cap_y, int64_t cap_z,
    int* __restrict__ lengths, int* __restrict__ err, int nstreams, int per_element_y) {
  const int lane = threadIdx.x & 63;
  const int sid = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  if (sid >= nstreams) return;
  // streams 0..B-1 are the long y strings, B..2B-1 the short z strings: the y waves share as few
  // workgroups (= CUs, which a persistent conv workgroup cannot use meanwhile) as possible
  const int nimg = nstreams >> 1;
  const int which = sid < nimg ? 1 : 0, b = which ? sid : sid - nimg;
  const int C = which ? M : N, HW = which ? HWy : HWz;
  const int64_t n = (int64_t)C * HW;
  const float* sym = which ? y + (size_t)b * n : z + (size_t)b * n;
  const bool per_element = which && per_element_y;  // spatial_params: one table row per y symbol
  const uint16_t* tab = (which ? tab_y + (size_t)b * (per_element ? (size_t)M * HWy : (size_t)M) * Lmax
                               : tab_z + (size_t)b * N * Lmax);
  const int smin = meta[4 * b + (which ? 0 : 2)], L = meta[4 * b + (which ? 1 : 3)];
  const int64_t stride = cap_z + cap_y;  // per image: [z bytes | y bytes], zero-initialised by the caller
  uint32_t* dst32 = (uint32_t*)(out + (size_t)b * stride + (which ? cap_z : 0));
  const int64_t cap_bits = (which ? cap_y : cap_z) * 8;
  if (L > Lmax || L < 1) {
    if (lane == 0) {
      atomicOr(err, 1);
      lengths[2 * b + which] = 0;
    }
    return;
  }

  // Two-deep software pipeline for the operand gather: symbols are loaded two groups ahead,
  // their table entries one group ahead, so neither dependent load round sits on the serial
  // chain even when L2 latency is several microseconds under a co-running conv kernel.
  auto sym_of = [&](int64_t g) -> float { return g < n ? sym[g] : 0.f; };
  auto pair_of = [&](float v, int64_t g) -> uint32_t {
    if (g >= n) return 0u;
    const int c = (int)(g / HW);
    int sc = (int)v - smin;
    if (sc < 0 || sc >= L) {  // cannot happen when meta came from dsic_latent_support on the same latents
      atomicOr(err, 2);
      sc = 0;
    }
    const uint16_t* t = tab + (size_t)(per_element ? g : (int64_t)c) * Lmax;
    const uint32_t c_low = t[sc];
    const uint32_t c_high = (sc == L - 1) ? 0x10000u : (uint32_t)t[sc + 1];
    return c_low | ((c_high - 1u) << 16);
  };

  uint32_t low = 0, high = 0xFFFFFFFFu, pending = 0;
  int64_t base_bits = 0;  // bits emitted so far (wave-uniform)
  int overflow = 0;

  uint32_t cur = pair_of(sym_of(lane), lane);
  float sym1 = sym_of(64 + lane);
  for (int64_t base = 0; base < n; base += 64) {
    const float sym2 = sym_of(base + 128 + lane);
    const uint32_t nxt = pair_of(sym1, base + 64 + lane);
    const int cnt = (int)((n - base) < 64 ? (n - base) : 64);
    uint32_t nbv, bitsv, pendv;   // lane j: final bits of symbol j, their count, inverse bits owed behind the first
    const uint32_t c_lo_v = cur & 0xFFFFu, c_hi_v = (cur >> 16) + 1u;
    if (cnt == 64 && !__any(c_hi_v == 0x10000u)) {
      // ---- fast path ----
      uint32_t rec_low = 0, rec_high = 0;
      uint32_t span = high - low + 1u, lowm1 = low - 1u;
      const uint32_t clo16 = c_lo_v << 16, chi16 = c_hi_v << 16;
      uint32_t cl = rlane<0>(clo16), ch = rlane<0>(chi16);
      enc_steps8<0>(low, lowm1, span, cl, ch, clo16, chi16, rec_low, rec_high);
      enc_steps8<8>(low, lowm1, span, cl, ch, clo16, chi16, rec_low, rec_high);
      enc_steps8<16>(low, lowm1, span, cl, ch, clo16, chi16, rec_low, rec_high);
      enc_steps8<24>(low, lowm1, span, cl, ch, clo16, chi16, rec_low, rec_high);
      enc_steps8<32>(low, lowm1, span, cl, ch, clo16, chi16, rec_low, rec_high);
      enc_steps8<40>(low, lowm1, span, cl, ch, clo16, chi16, rec_low, rec_high);
      enc_steps8<48>(low, lowm1, span, cl, ch, clo16, chi16, rec_low, rec_high);
      enc_steps8<56>(low, lowm1, span, cl, ch, clo16, chi16, rec_low, rec_high);
      high = low + span - 1u;
      // lanes: what symbol j emitted
      nbv = (uint32_t)__builtin_clz(rec_low ^ rec_high);
      bitsv = nbv ? rec_low >> (32u - nbv) : 0u;
      const uint32_t mv = (uint32_t)__builtin_clz((((rec_high | ~rec_low) << nbv) << 1) | 1u);
      // pending count in front of symbol j = sum of the E3 runs since the last symbol that
      // emitted bits (its own run included), or since the carry-in: a segmented prefix sum
      uint32_t T = mv;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(T, o, 64);
        if (lane >= o) T += up;
      }
      const uint32_t Tex = T - mv;
      const uint64_t heads = __ballot(nbv > 0);
      const uint64_t before = heads & ((1ull << lane) - 1ull);
      const int hb = 63 - __builtin_clzll(before | 1ull);
      const uint32_t Tex_h = __shfl(Tex, hb, 64);
      const uint32_t pend_in = before ? Tex - Tex_h : pending + Tex;
      pendv = nbv ? pend_in : 0u;
      const int hl = 63 - __builtin_clzll(heads | 1ull);
      const uint32_t T63 = __builtin_amdgcn_readlane(T, 63);
      const uint32_t Tex_hl = __builtin_amdgcn_readlane(Tex, hl);
      pending = heads ? T63 - Tex_hl : pending + T63;
    } else {
      // ---- general path (last partial group, or a symbol whose c_high is 65536) ----
      uint32_t rec0 = 0, rec1 = 0;  // lane j: (nb << 24 | bits) and the pending count flushed at symbol j
      for (int j = 0; j < cnt; ++j) {
        const uint32_t pr = __builtin_amdgcn_readlane(cur, j);
        const uint32_t c_low = pr & 0xFFFFu, c_high = (pr >> 16) + 1u;
        // span = high - low + 1 (up to 2^32): (span*c) >> 16 == (r*c + c) >> 16 with r = high - low
        const uint32_t r = high - low;
        const uint32_t hi_add = (uint32_t)(((uint64_t)r * c_high + c_high) >> 16);
        const uint32_t lo_add = (uint32_t)(((uint64_t)r * c_low + c_low) >> 16);
        high = (low - 1u) + hi_add;
        low = low + lo_add;
        // Between symbols high - low >= 2^30 (MSBs differ, no E3 pending) and every table interval
        // is >= 1/65536, so the new interval is >= 2^14 - 2 wide: low != high, nb <= 18.
        const int nb = __builtin_clz(low ^ high);             // E1/E2: leading bits now final
        const uint32_t bits = (low >> 1) >> (31 - nb);        // those nb bits (0 when nb == 0)
        const uint32_t flush = nb ? pending : 0u;             // inverse bits owed behind the first one
        pending = nb ? 0u : pending;
        // v_writelane has no clang builtin on this toolchain.  The lane select goes through M0
        // (an SGPR value plus an SGPR lane select would exceed the constant-bus limit); s_nop 3:
        // an SALU result needs 4 wait states before it is used as a lane select, and hipcc pads
        // nothing inside asm.  The kernel uses no LDS, GWS or movrel, so M0 is otherwise unused.
        const uint32_t w0 = bits | ((uint32_t)nb << 24);
        asm volatile("s_mov_b32 m0, %4\n\ts_nop 3\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %3, m0"
                     : "+v"(rec0), "+v"(rec1)
                     : "s"(w0), "s"(flush), "s"(j));
        low <<= nb;
        high = (high << nb) | ((1u << nb) - 1u);
        // E3: low = 01.., high = 10..: m consecutive (1,0) bit pairs below the MSB
        const uint32_t e3 = (low << 1) & ~(high << 1);        // bit 0 is 0, so ~e3 != 0
        const int m = __builtin_clz(~e3);
        pending += (uint32_t)m;
        low = (low << m) & 0x7FFFFFFFu;                       // MSB(low) is 0 here, so m == 0 is a no-op
        high = (high << m) | 0x80000000u | ((1u << m) - 1u);
      }
      nbv = rec0 >> 24;
      bitsv = rec0 & 0xFFFFFFu;
      pendv = rec1;
    }
    // parallel placement of this group's bits
    {
      const uint32_t len = nbv + pendv;
      uint32_t incl = len;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
      }
      const int64_t off = base_bits + (int64_t)(incl - len);
      if (nbv > 0) {
        if (pendv == 0) {
          put_bits(dst32, cap_bits, off, bitsv, (int)nbv, &overflow);
        } else {
          const uint32_t first = bitsv >> (nbv - 1);
          put_bits(dst32, cap_bits, off, first, 1, &overflow);
          if (!first) put_ones(dst32, cap_bits, off + 1, pendv, &overflow);   // inverse bits are 1s
          else if (off + 1 + (int64_t)pendv > cap_bits) overflow = 1;
          if (nbv > 1)
            put_bits(dst32, cap_bits, off + 1 + pendv, bitsv & ((1u << (nbv - 1)) - 1u), (int)nbv - 1, &overflow);
        }
      }
      base_bits += (int64_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    cur = nxt;
    sym1 = sym2;
  }
  // flush (torchac): one more pending bit, then the deciding bit and the pending run, zero padded
  pending += 1;
  const uint32_t bit = low < 0x40000000u ? 0u : 1u;
  if (lane == 0) {
    put_bits(dst32, cap_bits, base_bits, bit, 1, &overflow);
    if (!bit) put_ones(dst32, cap_bits, base_bits + 1, pending, &overflow);
    else if (base_bits + 1 + (int64_t)pending > cap_bits) overflow = 1;
  }
  const int64_t total_bits = base_bits + 1 + (int64_t)pending;
  if (__any(overflow)) {
    if (lane == 0) atomicOr(err, 4);
  }
  if (lane == 0) lengths[2 * b + which] = (int)((total_bits + 7) >> 3);
}

// Range decoder, one wave per stream (torchac decode_float_cdf, call sites :96,116).
//
// The reference computes count = ((value-low+1)*2^16 - 1) / span and searches the table for
// c[s] <= count < c[s+1].  For integers that is exactly  floor(span*c[s] / 2^16) <= value-low
// (no division), and the predicate is monotone in s, so all 64 lanes test one table entry each
// and a ballot + popcount gives s.  The interval update and renormalisation mirror the
// encoder (bulk shifts by clz); `value` takes the same shifts with fresh stream bits, and the
// m-step E3 correction  v <- 2(v - 2^30) + bit  collapses to flipping the top bit.  Stream
// bytes are fetched 256 at a time by the wave (zero past the end, like torchac's reader).
__global__ __launch_bounds__(256) void range_decode_kernel(const uint8_t* __restrict__ in,
                                                           int64_t stride,
                                                           const int* __restrict__ lengths, int lstride,
                                                           int loff, const int* __restrict__ meta,
                                                           int meta_off,
                                                           const uint16_t* __restrict__ tables, int Lmax,
                                                           int C, int HW, float* __restrict__ out,
                                                           int* __restrict__ err, int B,
                                                           int per_element) {
  const int lane = threadIdx.x & 63;
  const int b = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  if (b >= B) return;
  const int smin = meta[4 * b + meta_off], L = meta[4 * b + meta_off + 1];
  if (L > Lmax || L < 1) {
    if (lane == 0) atomicOr(err, 1);
    return;
  }
  const uint16_t* gt = tables + (size_t)b * (per_element ? (size_t)C * HW : (size_t)C) * Lmax;
  const uint32_t* src32 = (const uint32_t*)(in + (size_t)b * stride);  // stride is a multiple of 4
  const int nbytes = lengths[b * lstride + loff];
  const int ndw = (nbytes + 3) >> 2;
  const int64_t n = (int64_t)C * HW;
  float* dst = out + (size_t)b * n;

  // 64-dword window of the stream, big-endian words, bytes >= nbytes read as zero
  auto load_window = [&](int w0) -> uint32_t {
    const int k = w0 + lane;
    uint32_t v = 0;
    if (k < ndw) {
      v = __builtin_bswap32(src32[k]);
      const int valid = nbytes - 4 * k;  // 1..4 valid bytes in the last dword
      if (valid < 4) v &= 0xFFFFFFFFu << (8 * (4 - valid));
    }
    return v;
  };
  uint32_t win = load_window(0), win_next = load_window(64);
  int wi = 0;  // next dword to feed into the bit buffer
  auto next_dword = [&]() -> uint32_t {
    const uint32_t d = __builtin_amdgcn_readlane(win, wi & 63);
    ++wi;
    if ((wi & 63) == 0) {
      win = win_next;
      win_next = load_window(wi + 64);
    }
    return d;
  };
  uint64_t bitbuf = ((uint64_t)next_dword() << 32);
  bitbuf |= (uint64_t)next_dword();
  int avail = 64;
  auto take = [&](int k) -> uint32_t {  // next k (0..31) bits of the stream
    const uint32_t v = k ? (uint32_t)(bitbuf >> (64 - k)) : 0u;
    bitbuf <<= k;
    avail -= k;
    if (avail <= 32) {
      bitbuf |= (uint64_t)next_dword() << (32 - avail);
      avail += 32;
    }
    return v;
  };

  uint32_t low = 0, high = 0xFFFFFFFFu;
  uint32_t value = take(16);
  value = (value << 16) | take(16);

  const int nseg = (L + 63) >> 6;  // table entries per row, 64 per register
  if (nseg == 1 && !per_element) {
    // ---- fast path: the whole table row in one register, rows per channel ---------------------------------
    // Same arithmetic, shorter chain (round 3: 371 -> 188 ns per symbol at the bench shape): the state is (low, span) with span =
    // high - low + 1 kept mod 2^32 (0 means 2^32) and the lanes hold c << 16, so floor(span c / 2^16) is ONE
    // v_mul_hi_u32 per lane for the search and one s_mul_hi_u32 for each bound of the decoded symbol (the general
    // path multiplies in 64 bits); c_high = 65536 (last symbol) is hi_add = span.  The two renormalisation shifts
    // (E1/E2 by nb, E3 by m) are applied together and their stream bits taken together when nb + m < 32.  Decoded
    // symbols collect in a register (lane g & 63) and leave as one coalesced 256-byte store per 64 symbols
    // instead of a 4-byte store per symbol.  (Stream bits cut out of two window lanes by absolute bit position,
    // without the 64-bit buffer: +24 ns per symbol - two more v_readlane with an SGPR lane select.)
    const uint64_t valid = L >= 64 ? ~0ull : ((1ull << L) - 1ull);
    auto load_row16 = [&](int64_t rw) -> uint32_t {
      return lane < L ? ((uint32_t)gt[(size_t)rw * Lmax + lane]) << 16 : 0u;
    };
    uint32_t ck16 = load_row16(0), ck16_next = C > 1 ? load_row16(1) : 0u;
    uint32_t span = 0u, lowm1 = 0xFFFFFFFFu;   // low = 0
    int outv = 0;   // symbols (integers) of the current group of 64, one per lane
    int in_r = 0;
    int64_t rw = 0;
    for (int64_t g = 0; g < n; ++g) {
      const uint32_t d = value - low;
      const uint32_t bound = span ? __umulhi(span, ck16) : ck16;
      const int hits = __popcll(__ballot(bound <= d) & valid);   // >= 1: c[0] = 0
      // the decoded symbol's own bounds are two of the products the lanes just formed
      const uint32_t lo_add = __builtin_amdgcn_readlane(bound, hits - 1);
      const uint32_t hi_mul = __builtin_amdgcn_readlane(bound, hits & 63);
      const uint32_t hi_add = hits == L ? span : hi_mul;          // c_high = 65536: floor(span 2^16 / 2^16)
      {  // symbol g -> lane g & 63 of outv
        const int sv = hits - 1 + smin;
        const int ln = (int)(g & 63);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tv_writelane_b32 %0, %1, m0" : "+v"(outv) : "s"(sv), "s"(ln));
        if (ln == 63) dst[g - 63 + lane] = (float)outv;
      }
      const uint32_t low1 = low + lo_add;
      const uint32_t high1 = lowm1 + hi_add;
      const uint32_t span1 = hi_add - lo_add;
      const int nb = __builtin_clz(low1 ^ high1);
      const uint32_t q2 = (high1 | ~low1) << 1;
      const int m = __builtin_clz(q2 << nb);
      const int sh = nb + m;
      if (__builtin_expect(sh < 32, 1)) {
        span = span1 << sh;
        low = (low1 << sh) & 0x7FFFFFFFu;
        const uint32_t v2 = sh ? ((value << sh) | take(sh)) : value;
        value = m ? (v2 ^ 0x80000000u) : v2;
      } else {  // a long E3 run: the two shifts one after the other, as the general path
        uint32_t lo2 = low1 << nb, hi2 = (high1 << nb) | ((1u << nb) - 1u);
        value = nb ? ((value << nb) | take(nb)) : value;
        lo2 = (lo2 << m) & 0x7FFFFFFFu;
        hi2 = (hi2 << m) | 0x80000000u | ((1u << m) - 1u);
        value = ((value << m) | take(m)) ^ 0x80000000u;
        low = lo2;
        span = hi2 - lo2 + 1u;
      }
      lowm1 = low - 1u;
      if (++in_r == HW) {
        in_r = 0;
        rw += 1;
        ck16 = ck16_next;
        if (rw + 1 < C) ck16_next = load_row16(rw + 1);
      }
    }
    if (n & 63) {
      const int64_t g0 = n & ~(int64_t)63;
      if (lane < (int)(n & 63)) dst[g0 + lane] = (float)outv;
    }
    return;
  }
  // Table row of the current symbol: per channel (row changes every HW symbols) or per element
  // (spatial_params: a row per symbol).  Segment 0 of the NEXT row is prefetched while the
  // current symbol is decoded, so the row load never sits on the serial chain.
  // row index of symbol g, tracked incrementally (no 64-bit division on the serial chain)
  int64_t row = 0;   // row of the current symbol
  int in_row = 0;    // symbols already decoded from the current per-channel row
  auto load_seg = [&](const uint16_t* t, int seg) -> uint32_t {
    const int k = seg * 64 + lane;
    return k < L ? (uint32_t)t[k] : 0x10000u;  // entries past L act as c[L] = 65536
  };
  uint32_t ck0 = load_seg(gt, 0);
  for (int64_t g = 0; g < n; ++g) {
    {
      const uint16_t* t = gt + (size_t)row * Lmax;
      const bool new_row = per_element || in_row + 1 == HW;
      uint32_t ck0_next = ck0;
      if (new_row && g + 1 < n) ck0_next = load_seg(gt + (size_t)(row + 1) * Lmax, 0);
      const uint32_t d = value - low;
      const uint32_t r = high - low;  // span - 1
      // s = (number of k in [0,L) with floor(span*c[k]/2^16) <= d) - 1
      int cnt = 0;
      uint32_t c_low = 0, c_high = 0x10000u, carry_low = 0;
      bool carry = false;  // the previous 64-entry segment was entirely <= count
      for (int seg = 0; seg < nseg; ++seg) {
        const int k = seg * 64 + lane;
        const uint32_t ck = seg == 0 ? ck0 : load_seg(t, seg);
        const uint32_t bound = (uint32_t)(((uint64_t)r * ck + ck) >> 16);
        const int hits = __popcll(__ballot(k < L && bound <= d));
        if (hits == 0) {  // only after a full segment (c[0] = 0 always hits)
          if (carry) {
            c_low = carry_low;
            c_high = __builtin_amdgcn_readlane(ck, 0);
          }
          break;
        }
        cnt += hits;
        if (hits < 64) {
          c_low = __builtin_amdgcn_readlane(ck, hits - 1);
          c_high = __builtin_amdgcn_readlane(ck, hits);
          break;
        }
        carry = true;
        carry_low = __builtin_amdgcn_readlane(ck, 63);
        if (seg + 1 == nseg) {
          c_low = carry_low;
          c_high = 0x10000u;
        }
      }
      ck0 = ck0_next;
      if (new_row) {
        row += 1;
        in_row = 0;
      } else {
        in_row += 1;
      }
      const int sidx = cnt - 1;
      if (lane == 0) dst[g] = (float)(sidx + smin);
      // interval update + renormalisation (same arithmetic as the encoder)
      const uint32_t hi_add = (uint32_t)(((uint64_t)r * c_high + c_high) >> 16);
      const uint32_t lo_add = (uint32_t)(((uint64_t)r * c_low + c_low) >> 16);
      high = (low - 1u) + hi_add;
      low = low + lo_add;
      const int nb = __builtin_clz(low ^ high);
      low <<= nb;
      high = (high << nb) | ((1u << nb) - 1u);
      value = nb ? ((value << nb) | take(nb)) : value;
      const uint32_t e3 = (low << 1) & ~(high << 1);
      const int m = __builtin_clz(~e3);
      low = (low << m) & 0x7FFFFFFFu;
      high = (high << m) | 0x80000000u | ((1u << m) - 1u);
      value = m ? (((value << m) | take(m)) ^ 0x80000000u) : value;
    }
  }
}

}  // namespace dsic

using namespace dsic;

extern "C" int dsic_latent_support(const float* y_nchw, const float* z_nchw, int* meta, int B,
                                   int64_t n_y, int64_t n_z, int tail, void* stream) {
  DSIC_REQUIRE(y_nchw && z_nchw && meta, "latent_support: null pointer");
  DSIC_REQUIRE(B > 0 && n_y > 0 && n_z > 0 && tail >= 0, "latent_support: bad argument");
  hipLaunchKernelGGL(support_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, y_nchw, z_nchw, meta,
                     n_y, n_z, tail);
  return check_launch("latent_support");
}

// C counts table rows per image: channels, or channels*HW in per-element mode (sigma/nu then
// hold one value per latent element, NCHW order = symbol order).
static int tables_launch(bool student, const float* sigma, const float* nu, int per_image,
                         const int* meta, int meta_off, uint16_t* tables, int B, int C, int Lmax,
                         int* err, hipStream_t st) {
  DSIC_REQUIRE(sigma && meta && tables && err && (!student || nu), "cdf_tables: null pointer");
  DSIC_REQUIRE(B > 0 && C > 0 && Lmax >= 1 && Lmax <= 1000, "cdf_tables: Lmax=%d must be in [1,1000]", Lmax);
  const int ntables = B * C;
  const size_t shm = (size_t)4 * (2 * Lmax + 1) * sizeof(float);
  const int sb = per_image ? C : 0;
  if (student)
    hipLaunchKernelGGL(tables_kernel<true>, dim3(ceil_div(ntables, 4)), dim3(256), shm, st, sigma, nu, sb,
                       meta, meta_off, tables, C, Lmax, ntables, err);
  else
    hipLaunchKernelGGL(tables_kernel<false>, dim3(ceil_div(ntables, 4)), dim3(256), shm, st, sigma, nu, sb,
                       meta, meta_off, tables, C, Lmax, ntables, err);
  return check_launch("cdf_tables");
}

extern "C" int dsic_cdf_tables_gauss(const float* sigma_z, const int* meta, uint16_t* tables, int B,
                                     int N, int Lmax, int* err, void* stream) {
  return tables_launch(false, sigma_z, nullptr, 0, meta, 2, tables, B, N, Lmax, err, (hipStream_t)stream);
}

extern "C" int dsic_cdf_tables_student(const float* sigma, const float* nu, const int* meta,
                                       uint16_t* tables, int B, int rows, int Lmax, int* err,
                                       void* stream) {
  return tables_launch(true, sigma, nu, 1, meta, 0, tables, B, rows, Lmax, err, (hipStream_t)stream);
}

extern "C" int dsic_range_encode(const float* y_nchw, const float* z_nchw, const int* meta,
                                 const uint16_t* tab_y, const uint16_t* tab_z, int Lmax, int B, int M,
                                 int HWy, int N, int HWz, uint8_t* out, int64_t cap_y, int64_t cap_z,
                                 int* lengths, int* err, int streams_per_wg, int per_element_y,
                                 void* stream) {
  DSIC_REQUIRE(y_nchw && z_nchw && meta && tab_y && tab_z && out && lengths && err,
               "range_encode: null pointer");
  DSIC_REQUIRE(B > 0 && M > 0 && N > 0 && HWy > 0 && HWz > 0, "range_encode: empty latent");
  DSIC_REQUIRE(cap_y % 4 == 0 && cap_z % 4 == 0 && cap_y >= 8 && cap_z >= 8,
               "range_encode: capacities must be multiples of 4 and >= 8");
  DSIC_REQUIRE(streams_per_wg >= 1 && streams_per_wg <= 16, "range_encode: streams_per_wg must be in [1,16]");
  hipLaunchKernelGGL(range_encode_kernel, dim3(ceil_div(2 * B, streams_per_wg)), dim3(64 * streams_per_wg), 0,
                     (hipStream_t)stream,
                     y_nchw, z_nchw, meta, tab_y, tab_z, Lmax, M, HWy, N, HWz, out, cap_y, cap_z, lengths,
                     err, 2 * B, per_element_y ? 1 : 0);
  return check_launch("range_encode");
}

extern "C" int dsic_range_decode(const uint8_t* in, int64_t stride, const int* lengths, int lstride,
                                 int loff, const int* meta, int meta_off, const uint16_t* tables,
                                 int Lmax, int B, int C, int HW, int per_element, float* out_nchw,
                                 int* err, void* stream) {
  DSIC_REQUIRE(in && lengths && meta && tables && out_nchw && err, "range_decode: null pointer");
  DSIC_REQUIRE(B > 0 && C > 0 && HW > 0 && Lmax >= 1, "range_decode: bad argument");
  DSIC_REQUIRE(meta_off == 0 || meta_off == 2, "range_decode: meta_off must be 0 (y) or 2 (z)");
  DSIC_REQUIRE(stride % 4 == 0, "range_decode: stride must be a multiple of 4");
  // one wave per workgroup: a decoder wave has its CU's scalar unit to itself (as the encoder's waves)
  hipLaunchKernelGGL(range_decode_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, in, stride,
                     lengths, lstride, loff, meta, meta_off, tables, Lmax, C, HW, out_nchw, err, B, per_element ? 1 : 0);
  return check_launch("range_decode");
}

// A HIP stream restricted to a subset of the compute units (mask bit i = CU i
// enabled).  Lets the serial range coder own a few CUs while the conv kernels
// of the next batch run on the others.  The caller destroys it.
extern "C" int dsic_stream_create_masked(const uint32_t* mask_host, int words, void** stream_out) {
  DSIC_REQUIRE(mask_host && stream_out && words >= 1, "stream_create_masked: bad argument");
  hipStream_t st = nullptr;
  hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask_host);
  if (e != hipSuccess) {
    set_error("hipExtStreamCreateWithCUMask: %s", hipGetErrorString(e));
    return DSIC_EHIP;
  }
  *stream_out = (void*)st;
  return DSIC_OK;
}
extern "C" int dsic_stream_destroy(void* stream) {
  hipError_t e = hipStreamDestroy((hipStream_t)stream);
  if (e != hipSuccess) {
    set_error("hipStreamDestroy: %s", hipGetErrorString(e));
    return DSIC_EHIP;
  }
  return DSIC_OK;
}

// Host-side evaluation of the same table math (CPU tests compare it with the
// oracle without a GPU; the device kernels are compared on the GPU box).
extern "C" double dsic_host_normal_cdf(double x) { return dm::normal_cdf(x); }
extern "C" double dsic_host_student_t_cdf(double t, double nu) { return dm::student_t_cdf(t, nu); }
extern "C" float dsic_host_gaussian_cdf_f32(float x) { return dm::gaussian_cdf_f32(x); }
extern "C" float dsic_host_exp_f32(float x) { return (float)dm::exp((double)x); }
extern "C" int dsic_host_pmf_to_uint16_cdf(const float* pmf_host, int L, int C, uint16_t* out_host) {
  // :17-23 on a host pmf [L][C] (support axis first, like the reference) -> out [L+1][C]
  DSIC_REQUIRE(pmf_host && out_host && L >= 1 && C >= 1, "host_pmf_to_uint16_cdf: bad argument");
  for (int c = 0; c < C; ++c) {
    double cum = 0.0;
    out_host[c] = 0;
    for (int k = 0; k < L; ++k) {
      cum = cum + (double)pmf_host[(size_t)k * C + c];
      float v = (float)cum;
      if (k == L - 1 && v < 1.0f) v = 1.0f;
      float sc = v * 65535.0f;
      if (sc < 0.0f) sc = 0.0f;
      if (sc > 65535.0f) sc = 65535.0f;
      out_host[(size_t)(k + 1) * C + c] = (uint16_t)sc;
    }
  }
  return DSIC_OK;
}
extern "C" int dsic_host_cdf_table(int student, float sigma, float nu, int smin, int L,
                                   uint16_t* out_host, uint16_t* raw_host) {
  DSIC_REQUIRE(out_host && L >= 1 && L <= 4096, "host_cdf_table: bad argument");
  float F[4097], pmf[4096];
  for (int k = 0; k <= L; ++k)
    F[k] = student ? dm::table_cdf_student(smin, k, sigma, nu) : dm::table_cdf_gauss(smin, k, sigma);
  dm::finish_table(F, L, out_host, pmf, raw_host);
  return DSIC_OK;
}
