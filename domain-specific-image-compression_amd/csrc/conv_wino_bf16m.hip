// Split-bf16 Winograd F(2x2,3x3) for the LARGE layers: 64 Winograd tiles (16x16 output pixels) per workgroup,
// the 16 positions processed as TWO PASSES of 8 (code/modelv2/layers.py:54-72, 86-97: every conv(.,.,3,1),
// conv(C,C,5,2) over space-to-depth and ConvTranspose2d(.,.,5,2,2,1) whose H and W are multiples of 16).
//
// Why: conv_wino_bf16.hip (32 tiles x 16 positions x 128 channels in the accumulators) uses every fragment of the
// transformed weights U for ONE 32-tile MFMA operand: 128 KB of U per 16-channel chunk against 1536 cycles of MFMA
// issue per SIMD = 85 B/clk/CU from L2 - more than the vector-memory path delivers (64 B/clk/CU), and measured:
// with twice the MFMAs per fetched fragment (-DWB_ABL=128) that kernel takes 1.27-1.36x the time for 2x the matrix
// work.  Here a wave owns 4 positions x 2 M tiles x 32 channels (the same 128 accumulator registers) and uses every
// U fragment for both M tiles: half the U bytes per MFMA, and a fragment requested two steps ahead has 2 x 6 MFMAs
// to arrive instead of 2 x 3.
//
// Two passes.  Pass A accumulates the positions of the Winograd rows xi = 1, 2 over all input channels, pass B the
// rows xi = 0, 3.  Y = A^T M A with A^T = [[1,1,1,0],[0,1,-1,-1]]: row xi = 0 feeds only the output row i = 0, row
// xi = 3 only i = 1 (negated).  So the contribution P[i][j] of pass A is folded INTO pass B's accumulators before
// pass B starts - corner positions M[0][0] += P[0][0], M[0][3] -= P[0][1], M[3][0] -= P[1][0], M[3][3] += P[1][1] -
// and the fold after pass B yields the complete outputs: no partial sums leave the registers except the one
// exchange between the two waves of a channel group (LDS, 64 KB per tile at the pass boundary, 128 KB at the end).
//   pass A: wave (nt, pq) owns row xi = 1 + pq (4 positions); C[0] = M0+M1+M2, C[1] = M1-M2-M3 per row;
//           P[0][j] = C1[j] + C2[j], P[1][j] = C1[j] - C2[j]: pq 0 keeps column j = 0 (needs C2[0]), pq 1 column 1.
//   pass B: pq 0 owns (0,0) (0,1) (3,0) (3,1), pq 1 owns (0,2) (0,3) (3,2) (3,3); pq 0 finishes the outputs (i, 0),
//           pq 1 the outputs (i, 1), each with two accumulators of the other.
// The finished outputs (bias + GDN/IGDN/ReLU in registers, as in conv_wino_bf16.hip) leave as global_store_dword
// straight from the accumulator layout: lane = output channel, so 32 lanes write one pixel's 128 contiguous bytes.
// No output staging in LDS, no copy-out by the helper waves (they are the longer pole of a phase here).
//
// Helper waves, window staging, tile tickets, U stream layout and the V layout in LDS are those of
// conv_wino_bf16.hip with 64 tiles x 8 positions in place of 32 x 16; the same packed weights serve both kernels.
#include "conv_wino_bf16.h"

#ifndef WBM_STAMP
#define WBM_STAMP 0
#endif
#ifndef WBM_ABL
#define WBM_ABL 0   // diagnostic, wrong results: 1 every window load from tile (0,0) of image 0 (cache hits), 2 no window loads
#endif

namespace dsic {
namespace wbm {

using wb::Args;
using wb::bf16_hi;
using wb::bf16_lo;
using wb::bf16x8;
using wb::cvt_pk_bf16;
using wb::floatx16;
using wb::floatx2;
using wb::floatx4;
using wb::intx4;
using wb::Tile;
using wb::uintx2;
using wb::uintx4;

constexpr int P = 2;                            // bf16 planes
constexpr int CK = 16;                          // channels per chunk = one MFMA k-step
constexpr int NTILE = 64;                       // Winograd tiles per workgroup tile (8 x 8 = 16 x 16 pixels)
constexpr int ROWB = CK * 2;                    // bytes per (pos, tile) row of a plane
constexpr int POSB = NTILE * ROWB;              // 2048
constexpr int PLANEB = 8 * POSB;                // 8 positions per pass: 16384
constexpr int VBUFB = P * PLANEB;               // 32768
constexpr int XOFF = 2 * VBUFB;                 // exchange region A (region B is V buffer 1, idle during a fold)
constexpr int XBYTES = 8 * 4096;                // one floatx16 per MFMA wave
constexpr int SLOTOFF = XOFF + XBYTES;
constexpr int WINW = 18, WINH = 18;             // input window of a 16x16-pixel tile (halo 1)
constexpr int WINITEMS = WINW * WINH * 4;       // float4 items of one chunk: 1296
constexpr int WINB = WINITEMS * 16;             // 20736 bytes
constexpr int NLOAD = (WINITEMS + 255) / 256;   // 6 window loads per helper thread
constexpr int STAGEOFF = SLOTOFF + 64;
constexpr int NWIN = 3;                          // window buffers: one being transformed, two in flight (LDS-DMA)
constexpr int STAMPOFF = STAGEOFF + NWIN * WINB;
constexpr int PARAMOFF = STAMPOFF + (WBM_STAMP ? 1024 : 0);   // bias, beta, gamma: [3][128] floats
constexpr int LDS_TOTAL = PARAMOFF + 3 * 128 * 4;
constexpr int THREADS = 768;
constexpr int RING = 2;

static_assert(LDS_TOTAL <= 160 * 1024, "LDS budget");

#if WBM_STAMP
static __device__ long long wbm_stamps[256 * 256];   // low 32 bits of s_memtime (the LDS copy is 4 bytes per stamp)
#define MSTAMP(w, i)                                                                                   \
  do {                                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    if (lane == 0 && wave == (w) && tile_count == 3 && (i) >= 0 && (i) < 128)                          \
      ((unsigned*)(lds_raw + STAMPOFF))[((w) == 0 ? 0 : 128) + (i)] = (unsigned)__builtin_amdgcn_s_memtime();   \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
  } while (0)
#else
#define MSTAMP(w, i)
#endif

// The lane id, recomputed where it is used: volatile, so it is neither hoisted out of the tile loop nor kept in a
// register through the MFMA loops (which have 16 registers to spare beside accumulators, ring and V fragments).
__device__ __forceinline__ int lane_now() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

// Steps of a chunk for wave half PQ in pass PASS: 4 bits per step s, global position xi*4 + nu.  Positions that can
// be structurally zero come last (MODE 1: xi = 3 / nu = 3; MODE 2: xi = 0 / nu = 0), so the two fragments
// prefetched across a chunk boundary are live in (almost) every chunk.
template <int MODE, int PASS, int PQ>
struct Steps {
  // pass A: row xi = 1 + PQ; pass B: row xi = 0 (PQ 0) / 3 (PQ 1); nu ascending (MODE 2: descending, nu = 0 can vanish)
  static constexpr unsigned value =
      PASS == 0 ? (PQ == 0 ? (MODE == 2 ? 0x4567u : 0x7654u) : (MODE == 2 ? 0x89ABu : 0xBA98u))
                : (PQ == 0 ? (MODE == 2 ? 0x0123u : 0x3210u) : (MODE == 2 ? 0xCDEFu : 0xFEDCu));
};
template <int MODE, int PASS, int PQ>
__device__ __forceinline__ constexpr int gpos(int s) {
  return (int)((Steps<MODE, PASS, PQ>::value >> (4 * s)) & 15u);
}
// index of a global position inside the V buffer of its pass (8 positions: the pass's two xi rows x 4 nu)
template <int PASS>
__device__ __forceinline__ constexpr int lpos_of(int g) {
  return PASS == 0 ? ((g >> 2) - 1) * 4 + (g & 3) : ((g >> 2) == 3 ? 4 : 0) + (g & 3);
}
// step that holds global position g
template <int MODE, int PASS, int PQ>
__device__ __forceinline__ constexpr int step_of(int g) {
  for (int s = 0; s < 4; ++s)
    if (gpos<MODE, PASS, PQ>(s) == g) return s;
  return -1;
}

template <int MODE, bool NT_OUT>
__global__ __launch_bounds__(THREADS) void conv_wino_bf16m_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  float* const slots = (float*)(lds_raw + SLOTOFF);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  auto read_slot = [&](int s) {
    const intx4 v = *(const intx4*)(slots + 4 * s);
    Tile t;
    t.item = __builtin_amdgcn_readfirstlane(v[0]);
    t.tx = __builtin_amdgcn_readfirstlane(v[1]);
    t.ty = __builtin_amdgcn_readfirstlane(v[2]);
    t.n = __builtin_amdgcn_readfirstlane(v[3]);
    t.ks = 0;
    return t;
  };
  const int pshift = a.nphase == 4 ? 2 : 0;
  const int nchunks = a.Cin / CK;   // even, >= 4 (host)
  const int L = 2 * nchunks;        // chunk-passes of a tile: sigma < nchunks is pass A, the rest pass B
  int tile_count = 0;
  (void)tile_count;
  // per-channel epilogue parameters live in LDS: the MFMA waves read them at the fold (an LDS read is not ordered
  // behind their global stores, a register would be one of the 16 they have to spare)
  if (tid < 128) {
    float* const prm = (float*)(lds_raw + PARAMOFF);
    const bool cv = tid < a.Cout, gd = cv && (a.act == DSIC_ACT_GDN || a.act == DSIC_ACT_IGDN);
    prm[tid] = cv ? a.bias[tid] : 0.f;
    prm[128 + tid] = gd ? a.beta[tid] : 1.f;
    prm[256 + tid] = gd ? a.gamma[tid] : 0.f;
  }

  if (wave >= 8) {
    // =================================== helper waves ===========================================
    // thread = (Winograd tile pt of 64, channel quad pq of the 16-channel chunk): the 8 positions of the pass
    __builtin_amdgcn_s_setprio(3);
    const int ht = tid - 512;
    const int Cin = a.Cin;
    const int pt = ht >> 2, pq = ht & 3;
    const int ptx = pt & 7, pty = pt >> 3;
    const int vwrite = pt * ROWB + ((((pq >> 1) ^ ((pt >> 3) & 1))) << 4) + ((pq & 1) << 3);
    unsigned char* const vmine = lds_raw + vwrite;
    auto post = [&](int s, int item) {  // helper thread 0 only
      const int tile = item >> pshift;
      const int row = tile / a.tiles_x;
      const intx4 v = {item, tile - row * a.tiles_x, row % a.tiles_y, row / a.tiles_y};
      *(intx4*)(slots + 4 * s) = v;
    };
    // Input side.  The 18x18-pixel window of the tile (halo 1; 16 channels of the chunk, fp32) goes from global
    // memory straight into one of three LDS window buffers (buffer_load_dwordx4 ... lds: no registers, no staging
    // stores): item i = (pixel i >> 2, channel quad i & 3) lies at byte 16 i of the buffer, i.e. the 64 lanes of a
    // wave-instruction fill 1 KB at M0 + 16 lane.  Helper wave hw issues the items hw*64 + 256 j + lane (j < 6; the
    // last, partial piece belongs to the lanes 0..15 of wave 0).  The loads are inline asm: the compiler does not see
    // them (it would drain an LDS-DMA in front of every barrier), so the waits are counted here:
    //   phase sg: issue the window of chunk-pass sg + 3; transform sg + 1; wait for the window of sg + 2 (all but
    //             this wave's newest NDMA loads); barrier.
    // A window has two phases to arrive; an out-of-image pixel is an out-of-range offset (the DMA writes zeros).
    struct WinAim {
      unsigned off[NLOAD];
      uintx4 rsrc;
    };
    WinAim am;
    const int hw = wave - 8;
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)lds_raw;
    auto aim = [&](WinAim& m, const Tile& t0) {
      Tile t = t0;
      if (WBM_ABL & 1) { t.n = 0; t.tx = 0; t.ty = 0; }
      const unsigned long long base = (unsigned long long)(a.in + (size_t)t.n * a.H * a.W * Cin);
      m.rsrc = uintx4{(unsigned)base, (unsigned)(base >> 32) & 0xFFFFu, (unsigned)(a.H * a.W * Cin * 4), 0x00020000u};
#pragma unroll
      for (int j = 0; j < NLOAD; ++j) {
        const int i = ht + 256 * j;
        const int pix = i >> 2, q = i & 3;
        const int wy = pix / WINW, wx = pix - wy * WINW;
        const int gy = t.ty * 16 - 1 + wy, gx = t.tx * 16 - 1 + wx;
        const bool ok = i < WINITEMS && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && !(WBM_ABL & 2);
        m.off[j] = ok ? (unsigned)(((gy * a.W + gx) * Cin + 4 * q) * 4) : 0x80000000u;
      }
    };
    auto aim_nowhere = [&](WinAim& m) {
#pragma unroll
      for (int j = 0; j < NLOAD; ++j) m.off[j] = 0x80000000u;
    };
    auto dma = [&](int wbuf, int chunk) {
      const uintx4 rs = {(unsigned)__builtin_amdgcn_readfirstlane((int)am.rsrc[0]), (unsigned)__builtin_amdgcn_readfirstlane((int)am.rsrc[1]),
                         (unsigned)__builtin_amdgcn_readfirstlane((int)am.rsrc[2]), (unsigned)__builtin_amdgcn_readfirstlane((int)am.rsrc[3])};
      const unsigned soff = (unsigned)(chunk * (CK * 4));
#pragma unroll
      for (int j = 0; j < NLOAD; ++j) {
        const unsigned m0v = lds_base + (unsigned)(STAGEOFF + wbuf * WINB + 16 * (hw * 64 + 256 * j));
        if (j < NLOAD - 1 || ht < WINITEMS - 256 * (NLOAD - 1))   // the last piece: lanes 0..15 of helper wave 0
          asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                       :: "s"(m0v), "v"(am.off[j]), "s"(rs), "s"(soff) : "memory");
      }
    };
    // number of loads one dma() adds to this wave's vector-memory counter
    const bool last_piece = hw == 0;   // wave-uniform
    auto wait_all_but_newest = [&](int groups) {   // groups = 0, 1 or 2 windows may stay in flight
      if (groups == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (groups == 1) { if (last_piece) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); }
      else { if (last_piece) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); }
    };
    static_assert(NLOAD == 6, "the vmcnt immediates above count 6 (5) loads per window");
    // this thread's 4x4 patch inside a staged window: rows 2*pty .. +3, columns 2*ptx .. +3, quad pq
    const int patch0 = ((2 * pty) * WINW + 2 * ptx) * (CK * 4) + pq * 16;
    // a - b on a register pair: v_pk_add_f32 with the second operand negated (there is no v_pk_sub_f32, and the
    // compiler scalarises a vector fsub into v_sub_f32).  Every VALU instruction of a helper wave is paid in MFMA
    // time: other waves' vector instructions do not issue while an MFMA wave of the SIMD has MFMAs queued
    // (tools/coissue3.hip), so the transform's 72 subtractions per pass are issued as 36 packed ones.
    auto psub = [](floatx2 x, floatx2 y) {
      floatx2 r;
      asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
      return r;
    };
    auto sub4 = [&](floatx4 x, floatx4 y) {
      const floatx2 l = psub(floatx2{x[0], x[1]}, floatx2{y[0], y[1]}), h = psub(floatx2{x[2], x[3]}, floatx2{y[2], y[3]});
      return floatx4{l[0], l[1], h[0], h[1]};
    };
    auto split_store = [&](floatx4 v, unsigned char* dst) {
      const unsigned h0 = cvt_pk_bf16(v[0], v[1]), h1 = cvt_pk_bf16(v[2], v[3]);
      *(uintx2*)dst = uintx2{h0, h1};
      const floatx2 r01 = psub(floatx2{v[0], v[1]}, floatx2{bf16_lo(h0), bf16_hi(h0)});
      const floatx2 r23 = psub(floatx2{v[2], v[3]}, floatx2{bf16_lo(h1), bf16_hi(h1)});
      *(uintx2*)(dst + PLANEB) = uintx2{cvt_pk_bf16(r01[0], r01[1]), cvt_pk_bf16(r23[0], r23[1])};
    };
    // B^T d B for the two xi rows of a pass: window buffer wbuf -> V buffer vb.  pass A: xi 1 = r1 + r2, xi 2 = r2 - r1
    // (patch rows 1, 2 only); pass B: xi 0 = r0 - r2, xi 3 = r1 - r3.  zxi / znu: the structurally zero Winograd
    // row / column of this chunk (4 = none): never read by the MFMA waves, so neither transformed nor stored.
    auto commit = [&](int wbuf, int vb, bool passB, unsigned zxi, unsigned znu) {
      const unsigned char* src = lds_raw + STAGEOFF + wbuf * WINB + patch0;
      unsigned char* dst = vmine + vb * VBUFB;
      floatx4 xa[4], xb[4];
      bool a_live = true, b_live = true;
      if (!passB) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const floatx4 r1 = *(const floatx4*)(src + (1 * WINW + k) * (CK * 4));
          const floatx4 r2 = *(const floatx4*)(src + (2 * WINW + k) * (CK * 4));
          xa[k] = r1 + r2;
          xb[k] = sub4(r2, r1);
        }
      } else {
        a_live = zxi != 0u;
        b_live = zxi != 3u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const floatx4 r0 = *(const floatx4*)(src + (0 * WINW + k) * (CK * 4));
          const floatx4 r1 = *(const floatx4*)(src + (1 * WINW + k) * (CK * 4));
          const floatx4 r2 = *(const floatx4*)(src + (2 * WINW + k) * (CK * 4));
          const floatx4 r3 = *(const floatx4*)(src + (3 * WINW + k) * (CK * 4));
          xa[k] = sub4(r0, r2);
          xb[k] = sub4(r1, r3);
        }
      }
      // columns: nu 0: x0-x2, 1: x1+x2, 2: x2-x1, 3: x1-x3
      if (a_live) {
        if (znu != 0) split_store(sub4(xa[0], xa[2]), dst + 0 * POSB);
        split_store(xa[1] + xa[2], dst + 1 * POSB);
        split_store(sub4(xa[2], xa[1]), dst + 2 * POSB);
        if (znu != 3) split_store(sub4(xa[1], xa[3]), dst + 3 * POSB);
      }
      if (b_live) {
        if (znu != 0) split_store(sub4(xb[0], xb[2]), dst + 4 * POSB);
        split_store(xb[1] + xb[2], dst + 5 * POSB);
        split_store(sub4(xb[2], xb[1]), dst + 6 * POSB);
        if (znu != 3) split_store(sub4(xb[1], xb[3]), dst + 7 * POSB);
      }
    };
    auto zero_of = [&](const Tile& t, int k, unsigned& zxi, unsigned& znu) {
      zxi = 4;
      znu = 4;
      if (MODE == 1) {
        const int blk = k / (nchunks >> 2);
        if (blk >> 1) zxi = 3;
        if (blk & 1) znu = 3;
      } else if (MODE == 2) {
        const int phase = t.item & 3;
        if (phase >> 1) zxi = 0;
        if (phase & 1) znu = 0;
      }
    };
    auto next_ticket = [&]() { return (int)(atomicAdd(a.ticket, 1ULL) + gridDim.x); };

    if (ht == 0) {
      post(0, (int)blockIdx.x);
      post(1, next_ticket());
    }
    __syncthreads();  // P0
    Tile cur = read_slot(0);
    int ticket_pre = a.ntiles;
    aim(am, cur);
    dma(0, 0);
    dma(1, 1);
    dma(2, 2);
    if (ht == 0 && read_slot(1).item < a.ntiles) ticket_pre = next_ticket();
    wait_all_but_newest(2);
    __syncthreads();  // P1: the window of chunk 0 has landed (every helper wave has waited for its pieces)
    {
      unsigned zxi, znu;
      zero_of(cur, 0, zxi, znu);
      commit(0, 0, false, zxi, znu);    // V[0] = (cur, pass A, chunk 0)
    }
    wait_all_but_newest(1);             // window of chunk 1
    __syncthreads();  // P
    int wq = 1;                         // window buffer of the chunk-pass transformed in the coming phase
    int s_nxt = 1, s_wr = 2;
    while (cur.item < a.ntiles) {
      const Tile nxt = read_slot(s_nxt);
      const bool more = nxt.item < a.ntiles;
      tile_count++;
      // One phase (chunk-pass sg of this tile; targets past the last one belong to the next tile).  Every phase
      // issues the same memory operations whatever the tile: with no next tile the aim points nowhere.
      auto phase = [&](int sg) {
        MSTAMP(8, 4 * sg);
        {
          const int k3 = sg + 3;                                         // window of chunk-pass sg + 3
          if (k3 == L) {   // from here on every load is for the next tile
            if (more) aim(am, nxt); else aim_nowhere(am);
          }
          const int k3w = k3 < L ? k3 : k3 - L;
          dma(wq >= 1 ? wq - 1 : 2, k3w < nchunks ? k3w : k3w - nchunks);   // buffer (wq + 2) % 3
        }
        MSTAMP(8, 4 * sg + 1);
        {
          unsigned zxi, znu;                                             // target sg + 1
          const int tg = sg + 1;
          const bool passB = tg >= nchunks && tg < L;
          if (tg < L) zero_of(cur, passB ? tg - nchunks : tg, zxi, znu); else zero_of(nxt, 0, zxi, znu);
          commit(wq, tg & 1, passB, zxi, znu);
        }
        wq = wq == 2 ? 0 : wq + 1;
        wait_all_but_newest(1);                                          // window of chunk-pass sg + 2
        MSTAMP(8, 4 * sg + 2);
        __syncthreads();  // B_sg
        MSTAMP(8, 4 * sg + 3);
      };
      if (ht == 0 && more) post(s_wr, ticket_pre);
      for (int sg = 0; sg < L; sg += 2) {
        phase(sg);
        phase(sg + 1);
        if (sg + 2 == nchunks) {   // the MFMA waves fold pass A into pass B's accumulators (exchange through LDS,
          // region B = V buffer 1: not to be written before M4)
          __syncthreads();  // M1
          __syncthreads();  // M2
          __syncthreads();  // M3
          __syncthreads();  // M4
        }
      }
      if (ht == 0 && more && read_slot(s_wr).item < a.ntiles) ticket_pre = next_ticket();
      cur = nxt;
      const int s_old = s_nxt;
      s_nxt = s_wr;
      s_wr = s_old == 0 ? 2 : s_old - 1;
    }
#if WBM_STAMP
    if (wave == 8) {
      wbm_stamps[blockIdx.x * 256 + 128 + lane] = (long long)((unsigned*)(lds_raw + STAMPOFF))[128 + lane];
      wbm_stamps[blockIdx.x * 256 + 192 + lane] = (long long)((unsigned*)(lds_raw + STAMPOFF))[192 + lane];
    }
#endif
    if (ht == 0) {
      const unsigned long long done = atomicAdd(a.ticket + 1, 1ULL);
      if (done == (unsigned long long)gridDim.x - 1) {
        a.ticket[0] = 0ULL;
        a.ticket[1] = 0ULL;
      }
    }
    return;
  }

  // ===================================== MFMA waves ==============================================
  const int h = lane >> 5, l31 = lane & 31;
  const int nt = wave & 3, pqw = wave >> 2;
  const bool nvalid = nt * 32 < a.CoutP;
  constexpr bool ZSKIP = MODE != 0;
  // U stream: [phase][pos 16][chunk][plane][CoutP][16 bf16]; a fragment = 64 lanes x 16 bytes
  const unsigned plane_b = (unsigned)a.CoutP * 32u;
  const unsigned chunk_b = plane_b * (unsigned)P;
  const unsigned pos_b = chunk_b * (unsigned)nchunks;
  const unsigned ulane = (unsigned)((((nvalid ? nt : 0) * 32 + l31) * 2 + h) * 16);
  // V reads: row l31 (+ 32 m) of a position, k-block h (swizzled)
  const int aread = l31 * ROWB + ((h ^ ((l31 >> 3) & 1)) << 4);
  // exchange slots: [region][wave][quarter 4][lane][4 floats].  Addresses are rebuilt where they are used (from a
  // laundered lane id: the compiler would otherwise keep half a dozen tile-invariant address registers alive
  // through the MFMA loops, which have 16 registers to spare).
  auto xaddr = [&](bool region_b, bool partner) {
    const int ln = lane_now();
    return lds_raw + (region_b ? VBUFB : XOFF) + ((partner ? wave ^ 4 : wave) * 4096) + ln * 16;
  };
#define xa_mine xaddr(false, false)
#define xa_part xaddr(false, true)
#define xb_mine xaddr(true, false)
#define xb_part xaddr(true, true)
  auto xwrite = [&](unsigned char* p, const floatx16& v) {
#pragma unroll
    for (int q = 0; q < 4; ++q) *(floatx4*)(p + q * 1024) = floatx4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
  };
  auto xread = [&](const unsigned char* p) {
    floatx16 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const floatx4 x = *(const floatx4*)(p + q * 1024);
      v[4 * q] = x[0]; v[4 * q + 1] = x[1]; v[4 * q + 2] = x[2]; v[4 * q + 3] = x[3];
    }
    return v;
  };
  // output addressing (bytes): pixel of Winograd tile (ty, tx) of the workgroup tile, output (i, j) of that tile
  //   = tbase + ty*SY + tx*SX + i*SI + j*SJ + channel*4
  int SJ, SI;
  if (a.nphase == 4) {
    SJ = 2 * a.ostride * 4;
    SI = 2 * (2 * a.W) * a.ostride * 4;
  } else if (a.s2d) {
    SJ = a.Cout * 4;
    SI = 2 * a.Cout * 4;
  } else {
    SJ = a.ostride * 4;
    SI = a.W * a.ostride * 4;
  }
  const int SX = a.nphase == 4 ? 2 * SJ : a.s2d ? 4 * a.Cout * 4 : 2 * SJ;
  const int SY = a.nphase == 4 ? 2 * SI : a.s2d ? (a.W >> 1) * 4 * a.Cout * 4 : 2 * SI;

  auto mfma_waves = [&](auto pq_tag) {
  constexpr int PQ = decltype(pq_tag)::value;
  floatx16 acc[8];   // [step 4][m 2]
  bf16x8 Bq[RING][P];
  __syncthreads();  // P0
  __syncthreads();  // P1
  __syncthreads();  // P
  Tile cur = read_slot(0);
  const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)a.u, 0, (int)(16u * pos_b * (unsigned)a.nphase), 0x00020000);
  auto soff_item = [&](const Tile& t) { return (unsigned)(t.item & (a.nphase - 1)) * (unsigned)a.u_phase_bytes; };
  unsigned soff_phase = soff_item(cur);
  auto fetch = [&](bf16x8 (&dst)[P], unsigned so) {
#pragma unroll
    for (int q = 0; q < P; ++q)
      dst[q] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(urs, ulane, so + (unsigned)q * plane_b, 0));
  };
#pragma unroll
  for (int f = 0; f < RING; ++f) fetch(Bq[f], soff_phase + (unsigned)gpos<MODE, 0, PQ>(f) * pos_b);
  int s_nxt = 1;
  while (cur.item < a.ntiles) {
    tile_count++;
    const Tile nxt = read_slot(s_nxt);
    const unsigned soff_phase_nxt = soff_item(nxt.item < a.ntiles ? nxt : cur);
    unsigned tzero_xi = 4, tzero_nu = 4;   // MODE 2: the phase's zero row / column, for the whole tile
    if (MODE == 2) {
      const int phase = cur.item & 3;
      if (phase >> 1) tzero_xi = 0;
      if (phase & 1) tzero_nu = 0;
    }
    // one chunk of one pass: 4 steps x (U fragment pair from the ring; per M tile: V pair from LDS, 3 MFMAs)
    auto chunk_body = [&](auto pass_tag, auto first_tag, int chunk) {
      constexpr int PASS = decltype(pass_tag)::value;
      constexpr bool FIRST = decltype(first_tag)::value;
      const bool last = chunk + 1 == nchunks;
      MSTAMP(0, 3 * (PASS * nchunks + chunk));
      unsigned zero_xi = tzero_xi, zero_nu = tzero_nu;
      if (MODE == 1) {
        const int blk = chunk / (nchunks >> 2);
        zero_xi = (blk >> 1) ? 3 : 4;
        zero_nu = (blk & 1) ? 3 : 4;
      }
      auto is_zero = [&](int s) {
        const int g = gpos<MODE, PASS, PQ>(s);
        return (unsigned)(g >> 2) == zero_xi || (unsigned)(g & 3) == zero_nu;
      };
      const unsigned char* vb = lds_raw + ((PASS * nchunks + chunk) & 1) * VBUFB + aread;
      auto vaddr = [&](int sub, int q) {   // sub = 2*s + m
        return vb + lpos_of<PASS>(gpos<MODE, PASS, PQ>(sub >> 1)) * POSB + (sub & 1) * (32 * ROWB) + q * PLANEB;
      };
      bf16x8 Aq[P];
#pragma unroll
      for (int q = 0; q < P; ++q) Aq[q] = *(const bf16x8*)vaddr(0, q);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bool live = !ZSKIP || !is_zero(s);  // wave-uniform
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const int sub = 2 * s + m;
          if (live) {
            const floatx16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            floatx16 c = FIRST ? zero : acc[sub];
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Aq[0], Bq[s % RING][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Aq[1], Bq[s % RING][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Aq[0], Bq[s % RING][0], c, 0, 0, 0);
            acc[sub] = c;
          } else if (FIRST) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[sub][e] = 0.f;
          }
          // V of the next sub-step, issued right behind the MFMAs that read the current one
          if (sub + 1 < 8) {
#pragma unroll
            for (int q = 0; q < P; ++q) Aq[q] = *(const bf16x8*)vaddr(sub + 1, q);
          }
        }
        {
          // refill this ring slot with the fragments of step s + RING (the next chunk's first steps past 3).  Not
          // across a fold: the ring is refilled behind it, so that its 16 registers are free for the fold
          const int f = s + RING;
          const unsigned so = f < 4 ? soff_phase + (unsigned)gpos<MODE, PASS, PQ>(f) * pos_b + (unsigned)chunk * chunk_b
                                    : soff_phase + (unsigned)gpos<MODE, PASS, PQ>(f - 4) * pos_b + (unsigned)(chunk + 1) * chunk_b;
          if (f < 4 ? !(ZSKIP && is_zero(f)) : !last) fetch(Bq[s % RING], so);
        }
      }
      MSTAMP(0, 3 * (PASS * nchunks + chunk) + 1);
      __syncthreads();  // B_sigma
      MSTAMP(0, 3 * (PASS * nchunks + chunk) + 2);
    };

    // ---------------------------------- pass A ------------------------------------------------
    chunk_body(std::integral_constant<int, 0>{}, std::true_type{}, 0);
    for (int chunk = 1; chunk < nchunks; ++chunk) chunk_body(std::integral_constant<int, 0>{}, std::false_type{}, chunk);
    // ---- fold pass A into the corner positions of pass B ---------------------------------------
    {
      // own row xi = 1 + PQ: M[nu][m] = acc[2 * step(nu) + m]; C[0] = M0 + M1 + M2, C[1] = M1 - M2 - M3
      constexpr int XI = 1 + PQ;
      constexpr int s0 = step_of<MODE, 0, PQ>(XI * 4 + 0), s1 = step_of<MODE, 0, PQ>(XI * 4 + 1);
      constexpr int s2 = step_of<MODE, 0, PQ>(XI * 4 + 2), s3 = step_of<MODE, 0, PQ>(XI * 4 + 3);
      floatx16 c0[2], c1[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        c0[m] = (acc[2 * s0 + m] + acc[2 * s1 + m]) + acc[2 * s2 + m];
        c1[m] = (acc[2 * s1 + m] - acc[2 * s2 + m]) - acc[2 * s3 + m];
      }
      // pass B: this wave owns row xi = 0 (PQ 0) or 3 (PQ 1); its corner positions nu = 0, 3 start from
      //   (0,0) = P[0][0] = C1[0] + C2[0]    (0,3) = -P[0][1] = -(C1[1] + C2[1])
      //   (3,0) = -P[1][0] = C2[0] - C1[0]   (3,3) = P[1][1] = C1[1] - C2[1]          (C1: row 1 = PQ 0, C2: row 2 = PQ 1)
      constexpr int XB = PQ == 0 ? 0 : 3;
      constexpr int t0 = step_of<MODE, 1, PQ>(XB * 4 + 0), t1 = step_of<MODE, 1, PQ>(XB * 4 + 1);
      constexpr int t2 = step_of<MODE, 1, PQ>(XB * 4 + 2), t3 = step_of<MODE, 1, PQ>(XB * 4 + 3);
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        xwrite(xa_mine, c0[m]);
        xwrite(xb_mine, c1[m]);
        MSTAMP(0, 122 + 2 * m);
        __syncthreads();  // M1 / M3
        MSTAMP(0, 123 + 2 * m);
        const floatx16 g0 = xread(xa_part), g1 = xread(xb_part);
        if (PQ == 0) {
          acc[2 * t0 + m] = c0[m] + g0;
          acc[2 * t3 + m] = -(c1[m] + g1);
        } else {
          acc[2 * t0 + m] = c0[m] - g0;
          acc[2 * t3 + m] = g1 - c1[m];
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          acc[2 * t1 + m][e] = 0.f;
          acc[2 * t2 + m][e] = 0.f;
        }
        __syncthreads();  // M2 / M4: both regions have been read (M4: V buffer 1 belongs to the helpers again)
      }
      MSTAMP(0, 126);
    }
#pragma unroll
    for (int f = 0; f < RING; ++f) fetch(Bq[f], soff_phase + (unsigned)gpos<MODE, 1, PQ>(f) * pos_b);
    // ---------------------------------- pass B ------------------------------------------------
    for (int chunk = 0; chunk < nchunks; ++chunk) chunk_body(std::integral_constant<int, 1>{}, std::false_type{}, chunk);
    soff_phase = soff_phase_nxt;

    // ---- final fold: output row i = PQ from the wave's own row; bias + activation; stores --------
    // Y[0][0] = M00 + M01 + M02, Y[0][1] = M01 - M02 - M03 (row 0); Y[1][j] = -(the same of row 3).  No exchange and
    // no barrier: every wave transposes its outputs through its own 4 KB of region A (lane = channel -> lane = channel
    // quad of a tile) and stores 16 bytes per lane, 8 pixels x 128 bytes per instruction.
    MSTAMP(0, 120);
    {
      const int phase = cur.item & (a.nphase - 1);
      const int ppy = phase >> 1, ppx = phase & 1;
      const int OH = a.nphase == 4 ? 2 * a.H : a.H, OW = a.nphase == 4 ? 2 * a.W : a.W;
      const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(a.out + (size_t)cur.n * OH * OW * a.ostride + a.ooff), 0, (OH * OW * a.ostride - a.ooff) * 4, 0x00020000);
      const unsigned tbase =
          a.nphase == 4 ? (unsigned)(((32 * cur.ty + ppy) * OW + 32 * cur.tx + ppx) * a.ostride * 4)
          : a.s2d       ? (unsigned)(((8 * cur.ty) * (a.W >> 1) + 8 * cur.tx) * 4 * a.Cout * 4)
                        : (unsigned)((16 * cur.ty * a.W + 16 * cur.tx) * a.ostride * 4);
      constexpr int XB = PQ == 0 ? 0 : 3;
      constexpr int t0 = step_of<MODE, 1, PQ>(XB * 4 + 0), t1 = step_of<MODE, 1, PQ>(XB * 4 + 1);
      constexpr int t2 = step_of<MODE, 1, PQ>(XB * 4 + 2), t3 = step_of<MODE, 1, PQ>(XB * 4 + 3);
      // transposition area of this wave: [tile row 32][32 channels] floats
      const int ln = lane_now();   // (see xaddr)
      float* const tw = (float*)(lds_raw + XOFF + wave * 4096) + ln + 96 * (ln >> 5);   // lane (h, l31) -> row 4 h, channel l31: 128 h + l31; + row(e) * 32
      const float* const tr = (const float*)(lds_raw + XOFF + wave * 4096) + ln * 4;   // row lane >> 3, quad lane & 7: 32 (lane >> 3) + 4 (lane & 7) = 4 lane; + k * 256
      const float* const prm = (const float*)(lds_raw + PARAMOFF) + nt * 32 + (ln & 31);
      const float pbias = prm[0], pbeta = prm[128], pgamma = prm[256];
      const int cq = nt * 32 + (ln & 7) * 4;     // first channel of this lane's quad
      const unsigned svoff = cq < a.Cout ? (unsigned)((ln >> 3) * SX + cq * 4) : 0x80000000u;
      auto emit = [&](auto act_tag, int m, int j, floatx16 y) {
        constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
          floatx2 v = {y[e], y[e + 1]};
          v = v + floatx2{pbias, pbias};
          if (ACT == DSIC_ACT_GDN || ACT == DSIC_ACT_IGDN) {
            v = gdn_pair<ACT == DSIC_ACT_IGDN>(v, floatx2{pbeta, pbeta}, floatx2{pgamma, pgamma});
          } else if (ACT == DSIC_ACT_RELU) {
            v[0] = v[0] > 0.f ? v[0] : 0.f;
            v[1] = v[1] > 0.f ? v[1] : 0.f;
          }
          const float v0 = v[0], v1 = v[1];
          tw[((e & 3) + 8 * (e >> 2)) * 32] = v0;
          tw[(((e + 1) & 3) + 8 * ((e + 1) >> 2)) * 32] = v1;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads below must see this wave's writes
        // tile rows 8k .. 8k+7 = Winograd tiles (ty = 4 m + k, tx = lane >> 3).  All four reads first, into registers
        // of their own; the address goes into the vector offset (no SGPR offset) and the stores are followed by wait
        // states: on this hardware a 16-byte buffer store with an SGPR offset reads its data registers tens of cycles
        // after it has issued, and neither the hardware nor the compiler (which knows the hazard for immediate offsets
        // only) keeps the next instructions from overwriting them - measured: the first registers of the last store of
        // an emit took the next emit's first products in a quarter of the lanes, under load only.
        floatx4 o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = *(const floatx4*)(tr + k * 256);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned so = tbase + (unsigned)((4 * m + k) * SY + PQ * SI + j * SJ);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, o[k]), ors, svoff + so, 0, NT_OUT ? 2 : 0);
        }
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
      };
      auto emit_act = [&](int m, int j, const floatx16& y) {
        if (a.act == DSIC_ACT_GDN)
          emit(std::integral_constant<int, DSIC_ACT_GDN>{}, m, j, y);
        else if (a.act == DSIC_ACT_IGDN)
          emit(std::integral_constant<int, DSIC_ACT_IGDN>{}, m, j, y);
        else if (a.act == DSIC_ACT_RELU)
          emit(std::integral_constant<int, DSIC_ACT_RELU>{}, m, j, y);
        else
          emit(std::integral_constant<int, DSIC_ACT_NONE>{}, m, j, y);
      };
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const floatx16 y0 = (acc[2 * t0 + m] + acc[2 * t1 + m]) + acc[2 * t2 + m];
        const floatx16 y1 = (acc[2 * t1 + m] - acc[2 * t2 + m]) - acc[2 * t3 + m];
        emit_act(m, 0, PQ == 0 ? y0 : -y0);
        emit_act(m, 1, PQ == 0 ? y1 : -y1);
      }
    }
    MSTAMP(0, 121);
#pragma unroll
    for (int f = 0; f < RING; ++f) fetch(Bq[f], soff_phase + (unsigned)gpos<MODE, 0, PQ>(f) * pos_b);
    cur = nxt;
    s_nxt = s_nxt == 2 ? 0 : s_nxt + 1;
  }
#if WBM_STAMP
  if (wave == 0) {
    wbm_stamps[blockIdx.x * 256 + lane] = (long long)((unsigned*)(lds_raw + STAMPOFF))[lane];
    wbm_stamps[blockIdx.x * 256 + 64 + lane] = (long long)((unsigned*)(lds_raw + STAMPOFF))[64 + lane];
  }
#endif
  };
  if (pqw == 0)
    mfma_waves(std::integral_constant<int, 0>{});
  else
    mfma_waves(std::integral_constant<int, 1>{});
}

#undef xa_mine
#undef xa_part
#undef xb_mine
#undef xb_part

}  // namespace wbm
}  // namespace dsic

using namespace dsic;

// Which layers take this kernel: a function of the layer geometry only (never of the batch size: a patch's bits
// must not depend on the batch it is in).  H and W multiples of 16 (no ragged tiles: the stores are unmasked) and
// at least 4 work items per image (32x32 outputs; ConvTranspose2d: 16x16 inputs).
extern "C" int dsic_wino_bf16_m64(int H, int W, int Cin, int nphase) {
  static int mode = -1;   // DSIC_WINO_M64=0: never (A/B runs)
  if (mode < 0) {
    const char* e = getenv("DSIC_WINO_M64");
    mode = e ? atoi(e) : 1;
  }
  if (!mode) return 0;
  if (H <= 0 || W <= 0 || H % 16 || W % 16 || Cin < 64 || Cin % 32) return 0;
  const int min_items = mode > 1 ? mode : 4;
  return (H / 16) * (W / 16) * nphase >= min_items;
}

int dsic_wbm_launch(wb::Args& a, hipStream_t st) {
  a.tiles_x = a.W / 16;
  a.tiles_y = a.H / 16;
  const int64_t nt = (int64_t)a.tiles_x * a.tiles_y * a.B * a.nphase;
  DSIC_REQUIRE(nt < ((int64_t)1 << 31), "conv_wino_bf16m: too many tiles");
  DSIC_REQUIRE((a.Cin / wbm::CK) % 2 == 0 && a.Cin / wbm::CK >= 4, "conv_wino_bf16m: Cin=%d", a.Cin);
  if (a.ostride <= 0) a.ostride = a.Cout;
  a.ntiles = (int)nt;
  a.ksplit = 1;
  a.nt_out = (int64_t)a.B * a.H * a.W * a.Cout * 4 * (a.nphase == 4 ? 4 : 1) > (300ll << 20);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  if (dev < 0 || dev >= 64) dev = 0;
  static bool attr_set[64] = {};
  if (!attr_set[dev]) {
    const void* fns[6] = {(const void*)wbm::conv_wino_bf16m_kernel<0, false>, (const void*)wbm::conv_wino_bf16m_kernel<1, false>,
                          (const void*)wbm::conv_wino_bf16m_kernel<2, false>, (const void*)wbm::conv_wino_bf16m_kernel<0, true>,
                          (const void*)wbm::conv_wino_bf16m_kernel<1, true>,  (const void*)wbm::conv_wino_bf16m_kernel<2, true>};
    for (int i = 0; i < 6; ++i) {
      const hipError_t e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, wbm::LDS_TOTAL);
      if (e != hipSuccess) {
        set_error("conv_wino_bf16m: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return DSIC_EHIP;
      }
    }
    attr_set[dev] = true;
  }
  static int max_grid_dev[64] = {};
  if (max_grid_dev[dev] == 0) {   // one persistent workgroup per compute unit (DSIC_WINO_GRID: experiments)
    const char* g = getenv("DSIC_WINO_GRID");
    int n = g ? atoi(g) : 0;
    if (n < 1 || n > 1024) {
      if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
    }
    max_grid_dev[dev] = n;
  }
  const int grid = nt < max_grid_dev[dev] ? (int)nt : max_grid_dev[dev];
#define WBM_LAUNCH(M)                                                                                            \
  do {                                                                                                           \
    if (a.nt_out)                                                                                                \
      hipLaunchKernelGGL((wbm::conv_wino_bf16m_kernel<M, true>), dim3(grid), dim3(wbm::THREADS), wbm::LDS_TOTAL, st, a);  \
    else                                                                                                         \
      hipLaunchKernelGGL((wbm::conv_wino_bf16m_kernel<M, false>), dim3(grid), dim3(wbm::THREADS), wbm::LDS_TOTAL, st, a); \
  } while (0)
  if (a.s2d_in)
    WBM_LAUNCH(1);
  else if (a.nphase == 4)
    WBM_LAUNCH(2);
  else
    WBM_LAUNCH(0);
#undef WBM_LAUNCH
  return check_launch("conv_wino_bf16m");
}

#if WBM_STAMP
extern "C" int dsic_debug_wbm_stamps(long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(wbm::wbm_stamps), sizeof(long long) * 256 * 256) == hipSuccess ? 0 : 2;
}
#endif
