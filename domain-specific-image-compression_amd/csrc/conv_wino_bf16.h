// Launch arguments, LDS layout, position tables and the fold of the split-bf16 Winograd kernel (conv_wino_bf16.hip).
// (A header since the fused-role experiment of round 2, commit 5f523cb: a second kernel built on the same pieces.)
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "common.h"

#ifndef WB_PLANES
#define WB_PLANES 2
#endif
#ifndef WB_STAMP
#define WB_STAMP 0   // diagnostic: cycle stamps of MFMA wave 0 and helper wave 8 for one tile (tools/wb_stamps.py)
#endif
#ifndef WB_ABL
#define WB_ABL 0     // compile-time ablations (diagnostic, wrong results): 1 U from one hot line, 2 no input loads,
#endif               // 4 no transform / V stores, 8 no fold, 16 no MFMAs, 32 half of the MFMA waves' V reads.  (A run-time switch would put a branch
                     // around every MFMA cluster and cost the loop its scheduling.)
#ifndef WB_ADOUBLE
#define WB_ADOUBLE 0
#endif
#ifndef WB_HPRIO
#define WB_HPRIO 3
#endif
#ifndef WB_STAMP_TILE
#define WB_STAMP_TILE 8
#endif
#ifndef WB_STAMP_C0
#define WB_STAMP_C0 0   // first chunk that gets stamps (20 chunks of the MFMA wave, 15 of the helper wave fit)
#endif

namespace dsic {
namespace wb {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef int intx4 __attribute__((ext_vector_type(4)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
typedef unsigned uintx2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Args {
  const float* in;
  const void* u;  // bf16 planes [phase][16 pos][Cin/16][PLANES][CoutP][16]
  const float* bias;
  const float* beta;
  const float* gamma;
  float* out;
  int B, H, W, Cin, Cout, CoutP;
  int act;
  unsigned long long* ticket;
  int nphase;
  int64_t u_phase_bytes;
  int s2d_in, s2d;
  int tiles_x, tiles_y, ntiles;
  int nt_out;
  int ostride, ooff;  // output pixel stride and channel offset in floats (a Cout slice of a wider tensor)
  // split-K: a work item is (tile, ks); it walks kchunks = Cin/16/ksplit chunks from ks*kchunks and writes its
  // un-biased, un-activated sums into out + ks*part_stride (same addressing); splitk_reduce_kernel finishes
  int ksplit, kchunks;
  int64_t part_stride;
};

constexpr int P = WB_PLANES;
constexpr int CK = 16;                          // channels per chunk = one MFMA k-step
constexpr int ROWB = CK * 2;                    // bytes per (pos, tile) row of a plane
constexpr int POSB = 32 * ROWB;                 // bytes per position of a plane
constexpr int PLANEB = 16 * POSB;               // 16384
constexpr int VBUFB = P * PLANEB;               // bytes per V buffer
constexpr int WP = 36;                          // floats per (plane, tile) row of the output region
constexpr int YOFF = 2 * VBUFB;                 // byte offset of the output region
constexpr int YBYTES = 16 * 32 * WP * 4;        // 73728
constexpr int SLOTOFF = YOFF + YBYTES;
constexpr int WINW = 18, WINH = 10;              // input window of a 16x8-pixel tile (halo 1)
constexpr int WINB = WINW * WINH * CK * 4;      // fp32 window of one chunk: 11520 bytes
constexpr int STAGEOFF = SLOTOFF + 64;          // two window buffers
constexpr int STAMPOFF = STAGEOFF + 2 * WINB;
constexpr int LDS_TOTAL = STAMPOFF + (WB_STAMP ? 1024 : 0);
constexpr int THREADS = 768;
constexpr int RING = 2;                         // position-steps of U fragments in flight per MFMA wave

static_assert(LDS_TOTAL <= 160 * 1024, "LDS budget");

#if WB_STAMP
static __device__ long long wb_stamps[256 * 128];
#define WSTAMP(w, i)                                                                               \
  do {                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    if (lane == 0 && wave == (w) && tile_count == WB_STAMP_TILE && (i) < 64)                       \
      ((long long*)(lds_raw + STAMPOFF))[((w) == 0 ? 0 : 64) + (i)] = __builtin_amdgcn_s_memtime(); \
    __builtin_amdgcn_sched_barrier(0);                                                             \
  } while (0)
#define WSTAMPC(w, i, lim)                 \
  do {                                     \
    if ((i) >= 0 && (i) < (lim)) WSTAMP(w, i); \
  } while (0)
#else
#define WSTAMP(w, i)
#define WSTAMPC(w, i, lim)
#endif

__device__ __forceinline__ floatx2 pk_sub(floatx2 a, floatx2 b) {
  floatx2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ floatx4 sub4(floatx4 a, floatx4 b) { return a - b; }
__device__ __forceinline__ floatx16 sub16(floatx16 a, floatx16 b) {
  floatx16 r;
#pragma unroll
  for (int i = 0; i < 16; i += 2) {
    const floatx2 x = {a[i], a[i + 1]}, y = {b[i], b[i + 1]};
    const floatx2 d = pk_sub(x, y);
    r[i] = d[0];
    r[i + 1] = d[1];
  }
  return r;
}

// two floats -> packed bf16 pair (round to nearest even), and the pair back as two floats
typedef __bf16 wb_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  // the compiler's own v_cvt_pk_bf16_f32 (it schedules around an instruction it knows; behind inline asm it pads
  // every use with s_nop, and a helper wave's instruction slots are MFMA time)
  return __builtin_bit_cast(unsigned, __builtin_convertvector(floatx2{a, b}, wb_bf16x2));
}
__device__ __forceinline__ float bf16_lo(unsigned pk) { return __builtin_bit_cast(float, pk << 16); }
__device__ __forceinline__ float bf16_hi(unsigned pk) { return __builtin_bit_cast(float, pk & 0xFFFF0000u); }

struct Tile {
  int item, tx, ty, n, ks;
};

// Which 8 of the 16 Winograd positions (xi, nu) an MFMA wave of position half ph owns: the checkerboard
// (xi + nu) & 1 == ph.  A structurally zero row (xi) or column (nu) of a chunk - space-to-depth blocks, ConvTranspose
// phases - then costs both halves two positions each (a split by rows left one half with 8 live positions beside 4).
// 4 bits per step pi: xi*4 + nu, positions that can be structurally zero last (MODE 1: xi = 3 / nu = 3, MODE 2:
// xi = 0 / nu = 0), so that the two fragments prefetched across a chunk boundary are always live.
template <int MODE, int PH>
struct PosTab {
  static constexpr unsigned value = MODE == 2 ? (PH ? 0xC431EB96u : 0x820FDA75u) : (PH ? 0xECB39641u : 0xFD7A8520u);
};

// one output (i, j) of A^T M A restricted to the 8 positions of half PH: sum of +-acc[pi]
template <int MODE, int PH, int I, int J>
__device__ __forceinline__ floatx16 fold_partial(const floatx16 (&acc)[8]) {
  constexpr int AT[2][4] = {{1, 1, 1, 0}, {0, 1, -1, -1}};
  floatx16 s;
  bool first = true;
#pragma unroll
  for (int pi = 0; pi < 8; ++pi) {
    const unsigned g = (PosTab<MODE, PH>::value >> (4 * pi)) & 15u;
    const int coef = AT[I][g >> 2] * AT[J][g & 3];
    if (coef == 0) continue;
    if (first) {
      s = coef > 0 ? acc[pi] : -acc[pi];
      first = false;
    } else {
      s = coef > 0 ? s + acc[pi] : s - acc[pi];
    }
  }
  return s;
}

}  // namespace wb
}  // namespace dsic
