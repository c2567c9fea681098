// Winograd F(2x2,3x3) fp32 convolution for the 3x3 stride-1 layers
// (conv(N,N,3,1) at code/modelv2/layers.py:56,62,67,86,90,94,108,109) with the
// same fused bias + GDN/IGDN/ReLU epilogue as conv_igemm.hip.
//
// Y = A^T [ sum_c (G g G^T) o (B^T d B) ] A needs 16 multiplies per 2x2 output
// tile and channel pair instead of 36: 2.25x fewer MFMA flops for the layers
// that hold 35 % of the model's MACs, at fp32 accuracy (the transforms only add,
// subtract and halve).  It is the same exact-fp32 MFMA instruction; only the
// summation order differs from a direct convolution.
//
// Persistent workgroups (one per CU) of 768 threads over output tiles of 16x8 pixels (= 8x4
// Winograd tiles = the 32 rows of one MFMA M tile).  The 16 Winograd positions are 16 independent
// GEMMs  D_p[32 tiles][Cout] += V_p[32][Cin] U_p[Cin][Cout].
//   waves 0..7  (MFMA waves): wave w owns column tile nt = w&3 (32 output channels) and positions
//               8*(w>>2) .. +7: 8 accumulators of 32x32 = 128 VGPRs, two such waves per SIMD.  Their
//               loop reads V from LDS and U (transformed weights, packed [pos][Cin/8][CoutP][8], L2)
//               and issues MFMAs, nothing else.
//   waves 8..11 (helpers, one per SIMD): load the NHWC input (16 float4 per thread and 32-channel
//               chunk, one chunk ahead), apply B^T d B and write V into the double-buffered LDS image
//               [pos][tile][36]; after a tile they copy the finished 2x2 outputs (folded and activated
//               by the MFMA waves inside the V buffer that has just been consumed) to HBM, 16 bytes
//               per lane.
// See DESIGN.md section 3 for the reasons (in-order vmcnt, VALU issue under MFMA) and the numbers.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

#ifndef WINO_STAMP
#define WINO_STAMP 0
#endif
#ifndef WINO_ABL
#define WINO_ABL 0  // diagnostic ablations (wrong results): 1 no epilogue call, 2 no transform/V stores, 4 no input loads, 8 no fold
#endif
#ifndef WINO_STAMP_CHUNK0
#define WINO_STAMP_CHUNK0 0  // first of the four chunks whose MFMA phases are stamped
#endif
#ifndef WINO_STAMP_TILE
#define WINO_STAMP_TILE 16  // which tile of a workgroup is stamped
#endif

namespace dsic {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef int intx4 __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

// a - b on two / four / sixteen floats with packed fp32 instructions.  The compiler packs fp32
// additions (v_pk_add_f32) but leaves subtractions scalar; the negation is an operand modifier of
// the same instruction, so a - b costs the same single issue slot.  Every VALU issue slot matters
// here: a SIMD cannot issue VALU work of any wave while an MFMA is waiting for the matrix pipe.
__device__ __forceinline__ floatx2 pk_sub(floatx2 a, floatx2 b) {
  floatx2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ floatx4 sub4(floatx4 a, floatx4 b) {
  const floatx2 lo = pk_sub(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1));
  const floatx2 hi = pk_sub(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
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

struct WinoArgs {
  const float* in;
  const float* u;  // [16][Cin/8][CoutP][8]
  const float* bias;
  const float* beta;
  const float* gamma;
  float* out;
  int B, H, W, Cin, Cout, CoutP;
  int act;
  unsigned long long* ticket;  // [0] tiles handed out beyond the first round, [1] finished workgroups (zeroed
                               // once by the caller; the last workgroup of a launch zeroes it again)
  int nphase;  // 1, or 4: ConvTranspose2d(5,2,2,1) as four 3x3 sub-pixel phase convs sharing the input;
               // work item w = spatial tile*4 + phase, U of phase p at u + p*u_phase_stride, output pixel
               // (2*oy+py, 2*ox+px) of a [B,2H,2W,Cout] tensor
  int64_t u_phase_stride;
  int s2d_in;  // input is the space-to-depth image of a 5x5/s2 layer: Cin = 4*Cs, channel block (a,b)
  int s2d;  // store space-to-depth: [B,H/2,W/2,4*Cout] (feeds a 5x5/s2 layer run as 3x3 over 4*C)
  int tiles_x, tiles_y, ntiles;  // 16x8-pixel output tiles
  int nt_out;  // stream the output past the caches (set by the host for outputs beyond the MALL's size)
};

#if WINO_STAMP
__device__ long long wino_stamps[256 * 32];
// cycle stamps go to LDS (a global store would sit in the wave's in-order vmcnt queue and perturb
// what is measured) and are copied out when the workgroup ends
#define STAMP_AT(w, i)                                                                            \
  do {                                                                                            \
    __builtin_amdgcn_sched_barrier(0); /* s_memtime does not depend on the MFMAs around it */   \
    if (WINO_STAMP && lane == 0 && wave == (w) && tile_count == WINO_STAMP_TILE)                  \
      ((long long*)(lds + 2 * WBUF + 16))[i] = __builtin_amdgcn_s_memtime();                      \
    __builtin_amdgcn_sched_barrier(0);                                                            \
  } while (0)
#define STAMP(i) STAMP_AT(0, i)
#else
#define STAMP(i)
#endif

constexpr int WCK = 32;                    // channels per chunk
constexpr int WP = WCK + 4;                // LDS floats per (pos, tile) row
constexpr int WBUF = 16 * 32 * WP;         // floats per V buffer
constexpr int WLDS_BYTES = 2 * WBUF * 4;   // 147456
constexpr int WLDS_TOTAL = WLDS_BYTES + 64 + (WINO_STAMP ? 256 : 0);  // + three tile-descriptor slots (+ 32 stamps)
constexpr int WTHREADS = 768;              // 8 MFMA waves + 4 helper waves
#ifndef WINO_RING
#define WINO_RING 4                        // U fragments in flight per MFMA wave (2 or 4)
#endif
#ifndef WINO_NT_BYTES
#define WINO_NT_BYTES (300ll << 20)  // outputs larger than this are stored non-temporal (they cannot stay in the 256 MB MALL)
#endif
#ifndef WINO_CPRIO
#define WINO_CPRIO 1                       // wave priority of an MFMA wave inside an MFMA cluster
#endif
#ifndef WINO_HPRIO
#define WINO_HPRIO 0                       // wave priority of the helper waves
#endif

__device__ __forceinline__ void wg_barrier() { __syncthreads(); }

struct WinoTile {
  int item, tx, ty, n;  // work item (tile*nphase + phase) and its 16x8-pixel tile coordinates
};

// MODE 0: plain 3x3 layer.  MODE 1 (space-to-depth input) and MODE 2 (ConvTranspose2d phases) have
// structurally zero Winograd positions whose MFMA clusters are skipped; MODE 0 carries no test in
// the loop.  Steps run position-major (all four 8-channel groups of a position, then the next
// position), MODE 2 from position 7 down: the positions that can be zero then sit mostly at the
// end of a chunk, where the operand prefetch of the skipped steps runs into the chunk barrier
// instead of stalling the next live step (a skipped step takes no time, so the U fragments of the
// steps behind it have had no time to arrive).
//
// Wave specialisation.  Waves 0..7 ("MFMA waves") only read operands and issue MFMAs; waves 8..11
// ("helpers", one per SIMD) load the NHWC input, apply B^T d B and fill the other LDS buffer.
// vmcnt retires in order, so a wave that mixes HBM-latency input loads with the L2-latency U
// stream stalls its MFMAs behind the slowest input load; separate waves have separate counters.
// 12 waves = 3 per SIMD: the kernel must fit 168 VGPRs (128 of them accumulators).
// The MFMA waves fold their accumulators into the 2x2 outputs inside the free V buffer (bias and
// activation fused into the fold's second half) and go on to the next tile; the helpers copy the
// finished outputs to HBM from there, behind the next chunk's input loads.
// Every wave executes the same barrier sequence: P0, P, then per tile B_0..B_{n-1}, E1, E2.
//
// Tiles are handed out dynamically (first round = blockIdx.x, then a global ticket): a CU that is
// slowed down - e.g. by co-resident waves of another stream - simply takes fewer tiles.  Helper
// thread 0 fetches the ticket one tile ahead, splits it into (tx, ty, n) (the only integer
// divisions of the kernel) and posts the descriptor in a 3-slot LDS ring: slot k%3 = the k-th
// tile of this workgroup.
template <int MODE>
__global__ __launch_bounds__(WTHREADS) void conv_wino_kernel(const WinoArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* slots = lds + 2 * WBUF;  // written before a workgroup barrier, read after it
  auto read_slot = [&](int s) {
    const intx4 v = *(const intx4*)(slots + 4 * s);
    WinoTile t;
    t.item = __builtin_amdgcn_readfirstlane(v[0]);
    t.tx = __builtin_amdgcn_readfirstlane(v[1]);
    t.ty = __builtin_amdgcn_readfirstlane(v[2]);
    t.n = __builtin_amdgcn_readfirstlane(v[3]);
    return t;
  };
  const int pshift = a.nphase == 4 ? 2 : 0;

  if (wave >= 8) {
    // =================================== helper waves ===========================================
    // thread = (Winograd tile pt, channel quad pq): the whole 4x4 pixel patch, 16 float4 loads
    // (8 adjacent lanes = 128 contiguous bytes), all 16 positions.
    __builtin_amdgcn_s_setprio(WINO_HPRIO);
    const int ht = tid - 512;
    const int Cin = a.Cin;
    const int nchunks = Cin / WCK;
    const int pt = ht >> 3, pq = ht & 7;
    const int ptx = pt & 7, pty = pt >> 3;
    const int vwrite = pt * WP + 4 * pq;  // + pos*32*WP
    auto post = [&](int s, int item) {  // helper thread 0 only
      const int tile = item >> pshift;
      const int row = tile / a.tiles_x;
      const intx4 v = {item, tile - row * a.tiles_x, row % a.tiles_y, row / a.tiles_y};
      *(intx4*)(slots + 4 * s) = v;
    };
    // Input side, in three steps so that the loads of the next chunk can be put in flight before
    // the helper waits at a barrier (a helper only gets issue slots while the MFMA waves idle: the
    // SIMD issues one VALU-class instruction at a time and a pending MFMA holds the port):
    //   aim(tile)    per-thread byte offsets of the 16 patch pixels inside the image, once per tile
    //   issue(chunk) 16 float4 loads of the aimed tile's chunk
    //   commit(vbuf) B^T d B and the 16 stores into the V buffer
    // Input loads are raw buffer loads: descriptor (image base, image bytes) + one 32-bit byte offset
    // per patch pixel + the chunk as scalar offset, i.e. no address arithmetic per load, and a pixel
    // outside the image gets an offset beyond the descriptor's range, which the hardware reads as 0.
    struct Aim {
      unsigned off[16];  // byte offsets of the 16 patch pixels inside the image (+ this thread's channel quad)
      __amdgpu_buffer_rsrc_t rsrc;  // uniform
    };
    auto aim = [&](Aim& m, const WinoTile& t) {
      const int gy0 = t.ty * 8 + 2 * pty - 1, gx0 = t.tx * 16 + 2 * ptx - 1;
      m.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (size_t)t.n * a.H * a.W * Cin), 0,
                                                 a.H * a.W * Cin * 4, 0x00020000);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int gy = gy0 + i;
        const bool yok = gy >= 0 && gy < a.H;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int gx = gx0 + k;
          const bool ok = yok && gx >= 0 && gx < a.W;
          m.off[i * 4 + k] = ok ? (unsigned)(((gy * a.W + gx) * Cin + 4 * pq) * 4) : 0x80000000u;
        }
      }
    };
    auto issue = [&](floatx4 (&d)[16], const Aim& m, int chunk) {
      if (WINO_ABL & 4) return;
#pragma unroll
      for (int p = 0; p < 16; ++p)
        d[p] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(m.rsrc, m.off[p], chunk * (WCK * 4), 0));
    };
    auto commit = [&](floatx4 (&d)[16], const Aim& m, float* vbuf) {
      if (WINO_ABL & 2) return;
      // columns, in place: (w0,w1,w2,w3) = (d0-d2, d1+d2, d2-d1, d1-d3)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const floatx4 d0 = d[i * 4 + 0], d1 = d[i * 4 + 1], d2 = d[i * 4 + 2], d3 = d[i * 4 + 3];
        d[i * 4 + 0] = sub4(d0, d2);
        d[i * 4 + 1] = d1 + d2;
        d[i * 4 + 2] = sub4(d2, d1);
        d[i * 4 + 3] = sub4(d1, d3);
      }
      // rows, straight to LDS: xi0 = r0-r2, xi1 = r1+r2, xi2 = r2-r1, xi3 = r1-r3
      float* dst = vbuf + vwrite;
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) {
        const floatx4 r0 = d[0 * 4 + nu], r1 = d[1 * 4 + nu], r2 = d[2 * 4 + nu], r3 = d[3 * 4 + nu];
        *(floatx4*)(dst + (0 * 4 + nu) * 32 * WP) = sub4(r0, r2);
        *(floatx4*)(dst + (1 * 4 + nu) * 32 * WP) = r1 + r2;
        *(floatx4*)(dst + (2 * 4 + nu) * 32 * WP) = sub4(r2, r1);
        *(floatx4*)(dst + (3 * 4 + nu) * 32 * WP) = sub4(r1, r3);
      }
    };
    // Finished outputs of a tile (bias and activation already applied by the MFMA waves): LDS slots
    // -> global memory, 16 bytes per lane.  The V plane layout of Y (plane p = 4*(2i+j) + g holds,
    // for output pixel (i,j) of every Winograd tile, channels 32g..32g+31) means a helper thread
    // reads exactly the 16 float4 slots it will overwrite with the next V.
    // Two steps: aim_out(tile) computes, while the MFMA waves fold, a buffer descriptor of the output
    // image and one byte offset per output pixel (out of range for a pixel beyond the image: the
    // hardware drops that store); store_outputs() is then nothing but 16 LDS reads and 16 buffer
    // stores - no VALU work, so the copy runs beside the next tile's MFMAs.
    struct OutAim {
      unsigned po[4];
      __amdgpu_buffer_rsrc_t rs;
    };
    auto next_ticket = [&]() { return (int)(atomicAdd(a.ticket, 1ULL) + gridDim.x); };
    auto aim_out = [&](OutAim& o, const WinoTile& t) {
      const int phase = t.item & (a.nphase - 1);
      const int ppy = phase >> 1, ppx = phase & 1;  // sub-pixel phase placement (ConvTranspose2d)
      const int OH = a.nphase == 4 ? 2 * a.H : a.H, OW = a.nphase == 4 ? 2 * a.W : a.W;
      o.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)t.n * OH * OW * a.Cout), 0,
                                               OH * OW * a.Cout * 4, 0x00020000);  // host checks < 2 GiB
#pragma unroll
      for (int ij = 0; ij < 4; ++ij) {
        const int oy = t.ty * 8 + 2 * pty + (ij >> 1);
        const int ox = t.tx * 16 + 2 * ptx + (ij & 1);
        const unsigned po =
            (unsigned)(a.nphase == 4 ? ((2 * oy + ppy) * OW + (2 * ox + ppx)) * a.Cout
                       : a.s2d       ? ((oy >> 1) * (a.W >> 1) + (ox >> 1)) * (4 * a.Cout) + ((oy & 1) * 2 + (ox & 1)) * a.Cout
                                     : (oy * a.W + ox) * a.Cout) * 4u + 16u * pq;
        o.po[ij] = oy < a.H && ox < a.W ? po : 0x80000000u;
      }
    };
    auto store_outputs = [&](const OutAim& o, const float* yreg) {
      const float* src = yreg + vwrite;
      const int ngroups = (a.Cout + 31) >> 5;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (g >= ngroups) break;                                      // uniform
        if (g == ngroups - 1 && g * 32 + 4 * pq >= a.Cout) continue;  // partial last group (Cout % 32 != 0)
#pragma unroll
        for (int ij = 0; ij < 4; ++ij) {
          const floatx4 v = *(const floatx4*)(src + (ij * 4 + g) * 32 * WP);
          if (a.nt_out)  // uniform
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, v), o.rs, o.po[ij], g * 128, 2);
          else
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, v), o.rs, o.po[ij], g * 128, 0);
        }
      }
    };
    int tile_count = 0;
    (void)tile_count;
    if (ht == 0) {
      post(0, (int)blockIdx.x);  // grid <= ntiles
      post(1, next_ticket());
    }
    wg_barrier();  // P0: the first two descriptors are posted
    WinoTile cur = read_slot(0);
    // The patch in flight and its aim live across tiles: the aim moves on to the next tile two
    // chunks before the current one ends, and (layers with more than one chunk) the loads of the next
    // tile's target 0 are issued before the fold barriers of the current one, so their latency and
    // that of the ticket fetch pass while the MFMA waves fold.
    Aim m;
    floatx4 d[16];
    int ticket_pre = a.ntiles;  // ticket for the descriptor posted in the coming tile's chunk 0 (thread 0)
    aim(m, cur);
    issue(d, m, 0);
    commit(d, m, lds);
    if (nchunks > 1) {
      if (ht == 0 && read_slot(1).item < a.ntiles) ticket_pre = next_ticket();
      issue(d, m, 1);  // target 0 of the first tile
    }
    wg_barrier();  // P
    int buf = 0, s_nxt = 1, s_wr = 2;
    OutAim oa;
    oa.rs = m.rsrc;  // (defined on every path; aimed before the first fold)
#pragma unroll
    for (int ij = 0; ij < 4; ++ij) oa.po[ij] = 0x80000000u;
    bool have_y = false;  // finished outputs of the previous tile wait in V[buf^1], aimed by oa
    while (cur.item < a.ntiles) {
      const WinoTile nxt = read_slot(s_nxt);
      const bool more = nxt.item < a.ntiles;
      tile_count++;
      // chunk c: the MFMA waves consume V[buf]; the helpers fill V[buf^1] with target c =
      // (cur, c+1), or (nxt, 0) for the last chunk.  Re-aiming is VALU work and happens before the
      // barrier; the 16 buffer loads need no VALU and are issued right behind the barrier, where
      // they fly during the MFMA phase.
      auto pre_aim = [&](int chunk) {
        if (chunk + 2 == nchunks && more) aim(m, nxt);
      };
      auto post_issue = [&](int chunk) {
        if (chunk + 2 < nchunks)
          issue(d, m, chunk + 2);
        else if (chunk + 2 == nchunks && more)
          issue(d, m, 0);
      };
      {  // chunk 0
        float* vnext = lds + (buf ^ 1) * WBUF;
        const bool tgt0 = nchunks > 1 || more;
        int ticket = ticket_pre;
        if (nchunks == 1) {
          // single-chunk layer: target 0 is (nxt, 0).  Ticket first (its return is then the oldest
          // entry of this wave's in-order vmcnt queue), then the input loads, then the output copy:
          // the loads' latency passes during the copy and no load waits behind a store.
          if (ht == 0 && more) ticket = next_ticket();
          aim(m, more ? nxt : cur);
          issue(d, m, 0);
        }
        if (have_y && !(WINO_ABL & 1)) store_outputs(oa, vnext);
        if (tgt0) commit(d, m, vnext);
        if (ht == 0 && more) post(s_wr, ticket);
        pre_aim(0);
        wg_barrier();  // B_0
        post_issue(0);
        buf ^= 1;
      }
      for (int chunk = 1; chunk < nchunks; ++chunk) {
        float* vnext = lds + (buf ^ 1) * WBUF;
        if (chunk + 1 < nchunks || more) commit(d, m, vnext);
        pre_aim(chunk);
        wg_barrier();  // B_chunk
        post_issue(chunk);
        buf ^= 1;
      }
      if (nchunks > 1 && more) {
        // next tile's target 0 = (nxt, 1) (the aim is on nxt since chunk nchunks-2) and the ticket
        // for the descriptor it will post: both in flight across the fold barriers
        if (ht == 0 && read_slot(s_wr).item < a.ntiles) ticket_pre = next_ticket();
        issue(d, m, 1);
      }
      aim_out(oa, cur);  // VALU work, done while the MFMA waves fold
      wg_barrier();  // E1
      wg_barrier();  // E2: the finished outputs of this tile lie in V[buf^1]
      have_y = true;
      cur = nxt;
      const int s_old = s_nxt;
      s_nxt = s_wr;
      s_wr = s_old == 0 ? 2 : s_old - 1;  // ring 0,1,2: cur slot of the finished tile becomes writable
    }
    if (have_y) store_outputs(oa, lds + (buf ^ 1) * WBUF);  // outputs of the last tile
#if WINO_STAMP
    if (wave == 8 && lane < 32) wino_stamps[blockIdx.x * 32 + lane] = ((long long*)(lds + 2 * WBUF + 16))[lane];
#endif
    if (ht == 0) {  // last workgroup out re-arms the ticket for the next launch on this stream
      const unsigned long long done = atomicAdd(a.ticket + 1, 1ULL);
      if (done == (unsigned long long)gridDim.x - 1) {
        a.ticket[0] = 0ULL;
        a.ticket[1] = 0ULL;
      }
    }
    return;
  }

  // ===================================== MFMA waves ==============================================
  // Wave w owns column tile nt = w&3 (32 output channels) and positions 8*(w>>2) .. +7.
  const int h = lane >> 5, l31 = lane & 31;
  const int nt = wave & 3, ph = wave >> 2;
  const bool nvalid = nt * 32 < a.CoutP;
  const int nchunks = a.Cin / WCK;
  constexpr bool ZSKIP = MODE != 0;
  constexpr int PDIR = MODE == 2 ? -1 : 1;  // step it = pi*4 + sub works on position p = (PDIR > 0 ? pi : 7 - pi)
  // U stream: packed [pos][Cin/8][CoutP][8] floats; the fetch position advances by one 8-channel
  // group (step_s bytes) per fragment, after four of them to the next position, after 32 to the
  // first position of the next chunk.  Kept as one 32-bit lane offset against a uniform 64-bit
  // base, so no per-fragment address is hoisted.
  const unsigned step_s = (unsigned)a.CoutP * 32u;             // bytes per 8-channel group
  const unsigned step_p = (unsigned)(a.Cin >> 3) * step_s;    // bytes per position
  const unsigned step_pos = (unsigned)PDIR * step_p - 3u * step_s;    // (p, sub 3) -> (p + PDIR, sub 0)
  const unsigned step_chunk = step_s - 7u * (unsigned)PDIR * step_p;  // (last p, sub 3) -> (first p, next chunk)
  // The lane part of the address never changes; the stream position is a scalar (buffer load with
  // SGPR offset), so the MFMA loop spends no VALU instruction on addressing.
  const unsigned ulane = (unsigned)((((nvalid ? nt : 0) * 32 + l31) * 8 + 4 * h) * 4);
  const unsigned soff0 = (unsigned)(ph * 8 + (PDIR > 0 ? 0 : 7)) * step_p;  // first fragment of a tile
  const int aread = ((ph * 8) * 32 + l31) * WP + 4 * h;  // + p*32*WP + sub*8
  auto pos_of = [](int it) { return PDIR > 0 ? it >> 2 : 7 - (it >> 2); };
  // epilogue parameters of this lane's output channel
  float pbias = 0.f, pbeta = 1.f, pgamma = 0.f;
  {
    const int col = nt * 32 + l31;
    if (col < a.Cout) {
      pbias = a.bias[col];
      if (a.act == DSIC_ACT_GDN || a.act == DSIC_ACT_IGDN) {
        pbeta = a.beta[col];
        pgamma = a.gamma[col];
      }
    }
  }
  const int stamp_wave = 0;
  int tile_count = 0;
  (void)stamp_wave; (void)tile_count;
  constexpr int R = WINO_RING;
  static_assert(R == 2 || R == 4, "ring depth must divide the 32 steps of a chunk");

  floatx16 acc[8];
  floatx4 Bq[R];
  wg_barrier();  // P0
  wg_barrier();  // P
  WinoTile cur = read_slot(0);
  const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)a.u, 0, (int)(16u * step_p * (unsigned)a.nphase), 0x00020000);  // all phases' U (host checks < 2 GiB)
  const unsigned phase_bytes = (unsigned)a.u_phase_stride * 4u;
  unsigned soff = soff0 + (unsigned)(cur.item & (a.nphase - 1)) * phase_bytes;  // fetch stream position
  // skip: the step is a structurally zero position of this chunk - its MFMAs will not run, so the
  // load is pointed at the tile's first fragment (an L1/L2 hit) instead of pulling a line of zeros
  // through L2; the stream position advances all the same.
  auto fetch = [&](int f, bool skip = false) {  // fragment of step f (mod 32) of the stream
    const unsigned so = skip ? soff0 : soff;
    const floatx4 v = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(urs, ulane, so, 0));
    soff += (f & 31) == 31 ? step_chunk : ((f & 3) == 3 ? step_pos : step_s);
    return v;
  };
#pragma unroll
  for (int f = 0; f < R - 1; ++f) Bq[f] = fetch(f);
  int buf = 0, s_nxt = 1;
  while (cur.item < a.ntiles) {
    const WinoTile nxt = read_slot(s_nxt);
    const unsigned soff_nxt = soff0 + (unsigned)((nxt.item < a.ntiles ? nxt.item : cur.item) & (a.nphase - 1)) * phase_bytes;
    tile_count++;
    // FIRST: chunk 0 of a tile starts every accumulator from a zero C operand (inline constant), so
    // the 128 accumulator registers are never cleared by VALU moves
    auto chunk_body = [&](auto first_tag, int chunk) {
      constexpr bool FIRST = decltype(first_tag)::value;
      if (chunk >= WINO_STAMP_CHUNK0 && chunk < WINO_STAMP_CHUNK0 + 4) STAMP((chunk - WINO_STAMP_CHUNK0) * 4 + 0);
      const bool last = chunk + 1 == nchunks;
      // Structurally zero Winograd positions: a 3-tap filter with a zero end tap has a zero
      // transform component (G (g0,g1,0)^T)[3] = 0, (G (0,g1,g2)^T)[0] = 0.  For the space-to-depth
      // form of a 5x5/s2 kernel the phases a=1 / b=1 lack the last row / column (xi=3 / nu=3
      // vanish); for a ConvTranspose2d phase py=1 / px=1 lacks the first row / column (xi=0 /
      // nu=0 vanish).  Those MFMA clusters are skipped: 49 instead of 64 position-phase GEMMs.
      unsigned zero_xi = 4, zero_nu = 4;  // 4 = none
      if (MODE == 1) {
        const int blk = chunk / (nchunks >> 2);  // channel block (a,b) = (blk>>1, blk&1)
        if (blk >> 1) zero_xi = 3;
        if (blk & 1) zero_nu = 3;
      } else if (MODE == 2) {
        const int phase = cur.item & 3;
        if (phase >> 1) zero_xi = 0;
        if (phase & 1) zero_nu = 0;
      }
      auto is_zero = [&](int it) {
        const int p = pos_of(it);
        const unsigned xi = (unsigned)(ph * 2 + (p >> 2)), nu = (unsigned)(p & 3);
        return xi == zero_xi || nu == zero_nu;
      };
      const float* vb = lds + buf * WBUF + aread;
      floatx4 Aq[2];
      Aq[0] = *(const floatx4*)(vb + pos_of(0) * 32 * WP);
#pragma unroll
      for (int it = 0; it < 32; ++it) {  // it = pi*4 + sub
        {
          const int f = it + R - 1;  // U fragment to fetch now (continuous across chunks and tiles)
          if (f == 32 && last) soff = soff_nxt;  // the stream moves on to the next tile: its phase's U, first group
          // (the first R-1 steps of the next chunk are never zero positions: position 0, or 7 in MODE 2)
          Bq[f % R] = fetch(f, ZSKIP && f < 32 && is_zero(f));
        }
        if (it + 1 < 32) {
          const int p1 = pos_of(it + 1), s1 = (it + 1) & 3;
          Aq[(it + 1) & 1] = *(const floatx4*)(vb + p1 * 32 * WP + s1 * 8);
        }
        const int p0 = pos_of(it);
        if (!ZSKIP || !is_zero(it)) {  // wave-uniform
          if (WINO_CPRIO) __builtin_amdgcn_s_setprio(WINO_CPRIO);
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const floatx16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[p0] = __builtin_amdgcn_mfma_f32_32x32x2f32(Aq[it & 1][s], Bq[it % R][s],
                                                           FIRST && (it & 3) == 0 && s == 0 ? zero : acc[p0], 0, 0, 0);
          }
          if (WINO_CPRIO) __builtin_amdgcn_s_setprio(0);
        } else if (FIRST && (it & 3) == 0) {  // a skipped position still has to read as zero in the fold
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[p0][e] = 0.f;
        }
      }
      if (chunk >= WINO_STAMP_CHUNK0 && chunk < WINO_STAMP_CHUNK0 + 4) STAMP((chunk - WINO_STAMP_CHUNK0) * 4 + 1);
      if (chunk >= WINO_STAMP_CHUNK0 && chunk < WINO_STAMP_CHUNK0 + 4) STAMP((chunk - WINO_STAMP_CHUNK0) * 4 + 2);
      wg_barrier();  // B_chunk
      if (chunk >= WINO_STAMP_CHUNK0 && chunk < WINO_STAMP_CHUNK0 + 4) STAMP((chunk - WINO_STAMP_CHUNK0) * 4 + 3);
      buf ^= 1;
    };
    chunk_body(std::true_type{}, 0);
    for (int chunk = 1; chunk < nchunks; ++chunk) chunk_body(std::false_type{}, chunk);


    // ---- inverse transform ------------------------------------------------------------------
    // Wave (nt, ph) holds M[xi][nu] for xi in {2ph, 2ph+1}.  N[xi][j] = (M A)[xi][j]:
    //   N[.][0] = M0 + M1 + M2,  N[.][1] = M1 - M2 - M3.
    // Y[i][j] = (A^T N)[i][j]:  Y[0] = N0 + N1 + N2,  Y[1] = N1 - N2 - N3.
    //   ph=0: own = N0 + N1 (row 0), sends N1 to row 1;   ph=1: own = -(N2 + N3) (row 1), sends N2 to row 0.
    // Each wave stores its own row's partial sum into the Y planes of the free V buffer, and after
    // a barrier adds the term it owes to the other row in place (read, add, write: after the
    // barrier exactly one lane touches each element): Y = own + received.
    STAMP(24);
    if (WINO_ABL & 8) {
      wg_barrier();
    } else {
      float* yreg = lds + (buf ^ 1) * WBUF;
      // element e of this lane: Winograd tile (e&3) + 8*(e>>2) + 4h, channel 32nt + l31
      float* yown = yreg + ((4 * (2 * ph) + nt) * 32 + 4 * h) * WP + l31;
      float* yoth = yreg + ((4 * (2 * (ph ^ 1)) + nt) * 32 + 4 * h) * WP + l31;
      floatx16 send[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const floatx16 na = j == 0 ? (acc[0] + acc[1]) + acc[2] : sub16(sub16(acc[1], acc[2]), acc[3]);  // local xi 0
        const floatx16 nb = j == 0 ? (acc[4] + acc[5]) + acc[6] : sub16(sub16(acc[5], acc[6]), acc[7]);  // local xi 1
        const floatx16 own = ph == 0 ? na + nb : -(na + nb);
        send[j] = ph == 0 ? nb : na;
#pragma unroll
        for (int e = 0; e < 16; ++e) yown[(4 * j * 32 + (e & 3) + 8 * (e >> 2)) * WP] = own[e];
      }
      wg_barrier();  // E1
      STAMP(25);
      // The wave that adds the second term holds the finished sum in registers, so bias and
      // activation are applied right here (this lane = one output channel: its three parameters
      // live in registers) and the helpers only copy the result out.
      auto finish_rows = [&](auto act_tag) {
        constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          floatx16 got;
#pragma unroll
          for (int e = 0; e < 16; ++e) got[e] = yoth[(4 * j * 32 + (e & 3) + 8 * (e >> 2)) * WP];
#pragma unroll
          for (int e = 0; e < 16; e += 2) {
            floatx2 v = {got[e], got[e + 1]};
            v = v + floatx2{send[j][e], send[j][e + 1]};
            v = v + floatx2{pbias, pbias};
            if (ACT == DSIC_ACT_GDN || ACT == DSIC_ACT_IGDN) {
              v = gdn_pair<ACT == DSIC_ACT_IGDN>(v, floatx2{pbeta, pbeta}, floatx2{pgamma, pgamma});
            } else if (ACT == DSIC_ACT_RELU) {
              v[0] = v[0] > 0.f ? v[0] : 0.f;
              v[1] = v[1] > 0.f ? v[1] : 0.f;
            }
            yoth[(4 * j * 32 + (e & 3) + 8 * (e >> 2)) * WP] = v[0];
            yoth[(4 * j * 32 + ((e + 1) & 3) + 8 * ((e + 1) >> 2)) * WP] = v[1];
          }
        }
      };
      if (a.act == DSIC_ACT_GDN)
        finish_rows(std::integral_constant<int, DSIC_ACT_GDN>{});
      else if (a.act == DSIC_ACT_IGDN)
        finish_rows(std::integral_constant<int, DSIC_ACT_IGDN>{});
      else if (a.act == DSIC_ACT_RELU)
        finish_rows(std::integral_constant<int, DSIC_ACT_RELU>{});
      else
        finish_rows(std::integral_constant<int, DSIC_ACT_NONE>{});
    }
    STAMP(26);
    wg_barrier();  // E2: the helpers copy the finished outputs from here and refill the buffer
    STAMP(27);
    cur = nxt;
    s_nxt = s_nxt == 2 ? 0 : s_nxt + 1;
  }
}

// U_p = G g G^T per (cout, cin), packed [16][Cin/8][CoutP][8].
// s2 = 0: g is the 3x3 kernel of a stride-1 conv, w [Cout][Cin][3][3].
// s2 = 1: w [Cout][Cs][5][5] is a 5x5 stride-2 pad-2 kernel; over the space-to-depth input
//         (channel c' = (a*2+b)*Cs + c holds x[2i+a][2j+b][c]) it is the 3x3 stride-1 kernel
//         g_ab[u][v] = w[2u+a][2v+b] (zero where 2u+a or 2v+b exceeds 4): Cin = 4*Cs.
__global__ void pack_wino_weight_kernel(const float* __restrict__ w, float* __restrict__ dst, int Cout,
                                        int Cin, int Cin8, int CoutP, int s2, int64_t total) {
  // s2 = 2..5: phase (s2-2) = py*2+px of ConvTranspose2d(5,2,2,1), w [Cin][Cout][5][5]:
  //   g[wr][wc] = w[c][n][py+4-2wr][px+4-2wc] for wr >= py, wc >= px, else 0
  //   (output (2i+py, 2j+px) reads input (i-1+wr, j-1+wc); from oy = 2*iy - 2 + ky).
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int j = i & 7;
  int64_t r = i >> 3;
  const int n = r % CoutP;
  r /= CoutP;
  const int c8 = r % Cin8;
  const int p = r / Cin8;
  const int xi = p >> 2, nu = p & 3;
  const int c = c8 * 8 + j;
  float v = 0.f;
  if (n < Cout && c < Cin) {
    float g[9];
    if (s2 >= 2) {
      const int py = (s2 - 2) >> 1, px = (s2 - 2) & 1;
      const float* gp = w + ((size_t)c * Cout + n) * 25;
#pragma unroll
      for (int wr = 0; wr < 3; ++wr)
#pragma unroll
        for (int wc = 0; wc < 3; ++wc)
          g[wr * 3 + wc] = (wr >= py && wc >= px) ? gp[(py + 4 - 2 * wr) * 5 + (px + 4 - 2 * wc)] : 0.f;
    } else if (!s2) {
      const float* gp = w + ((size_t)n * Cin + c) * 9;
#pragma unroll
      for (int t = 0; t < 9; ++t) g[t] = gp[t];
    } else {
      const int Cs = Cin >> 2, ab = c / Cs, cs = c % Cs;
      const int pa = ab >> 1, pb = ab & 1;
      const float* gp = w + ((size_t)n * Cs + cs) * 25;
#pragma unroll
      for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int v = 0; v < 3; ++v) {
          const int ky = 2 * u + pa, kx = 2 * v + pb;
          g[u * 3 + v] = (ky < 5 && kx < 5) ? gp[ky * 5 + kx] : 0.f;
        }
    }
    // row combination (G g)[xi][kx], then column combination with G^T
    float row[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const float g0 = g[0 * 3 + kx], g1 = g[1 * 3 + kx], g2 = g[2 * 3 + kx];
      row[kx] = xi == 0 ? g0 : (xi == 1 ? 0.5f * ((g0 + g1) + g2) : (xi == 2 ? 0.5f * ((g0 - g1) + g2) : g2));
    }
    v = nu == 0 ? row[0]
                : (nu == 1 ? 0.5f * ((row[0] + row[1]) + row[2])
                           : (nu == 2 ? 0.5f * ((row[0] - row[1]) + row[2]) : row[2]));
  }
  dst[i] = v;
}

}  // namespace dsic

using namespace dsic;

#if WINO_STAMP
extern "C" int dsic_debug_wino_stamps(long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(wino_stamps), sizeof(long long) * 256 * 32) == hipSuccess ? 0 : 2;
}
#endif

extern "C" int64_t dsic_wino_weight_floats(int Cout, int Cin) {
  return (int64_t)16 * (round_up(Cin, 8) / 8) * round_up(Cout, 32) * 8;
}

extern "C" int dsic_pack_wino_weight(const float* w_oihw, float* dst, int Cout, int Cin, void* stream) {
  DSIC_REQUIRE(w_oihw && dst && Cout > 0 && Cin > 0, "pack_wino_weight: bad argument");
  const int Cin8 = round_up(Cin, 8) / 8, CoutP = round_up(Cout, 32);
  const int64_t total = dsic_wino_weight_floats(Cout, Cin);
  hipLaunchKernelGGL(pack_wino_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, w_oihw, dst, Cout, Cin, Cin8, CoutP, 0, total);
  return check_launch("pack_wino_weight");
}

extern "C" int dsic_pack_wino_s2_weight(const float* w_oihw5, float* dst, int Cout, int Cs, void* stream) {
  DSIC_REQUIRE(w_oihw5 && dst && Cout > 0 && Cs > 0 && Cs % 8 == 0, "pack_wino_s2_weight: bad argument");
  const int Cin = 4 * Cs, Cin8 = Cin / 8, CoutP = round_up(Cout, 32);
  const int64_t total = dsic_wino_weight_floats(Cout, Cin);
  hipLaunchKernelGGL(pack_wino_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, w_oihw5, dst, Cout, Cin, Cin8, CoutP, 1, total);
  return check_launch("pack_wino_s2_weight");
}

extern "C" int dsic_pack_wino_convT_weight(const float* w_iohw5, float* dst, int Cin, int Cout, void* stream) {
  DSIC_REQUIRE(w_iohw5 && dst && Cout > 0 && Cin > 0 && Cin % 8 == 0, "pack_wino_convT_weight: bad argument");
  const int Cin8 = Cin / 8, CoutP = round_up(Cout, 32);
  const int64_t total = dsic_wino_weight_floats(Cout, Cin);
  for (int phase = 0; phase < 4; ++phase) {
    hipLaunchKernelGGL(pack_wino_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, w_iohw5, dst + (size_t)phase * total, Cout, Cin, Cin8, CoutP, 2 + phase,
                       total);
  }
  return check_launch("pack_wino_convT_weight");
}

static int wino_launch(WinoArgs& a, hipStream_t st);

extern "C" int dsic_conv_transpose2d_wino_nhwc(const float* in, const float* u_packed4, const float* bias,
                                               const float* beta, const float* gamma, float* out, int B,
                                               int H, int W, int Cin, int Cout, int act, void* ticket,
                                               void* stream) {
  DSIC_REQUIRE(in && u_packed4 && bias && out && ticket, "convT_wino: null pointer");
  DSIC_REQUIRE(B > 0 && H > 0 && W > 0, "convT_wino: empty tensor");
  DSIC_REQUIRE(Cin > 0 && Cin % 32 == 0, "convT_wino: Cin=%d must be a positive multiple of 32", Cin);
  DSIC_REQUIRE(Cout > 0 && Cout % 4 == 0 && Cout <= 128, "convT_wino: Cout=%d must be a multiple of 4, <= 128", Cout);
  DSIC_REQUIRE(act >= 0 && act <= 3, "convT_wino: act=%d", act);
  DSIC_REQUIRE(!(act == DSIC_ACT_GDN || act == DSIC_ACT_IGDN) || (beta && gamma), "convT_wino: IGDN needs beta and gamma");
  WinoArgs a{};
  a.in = in; a.u = u_packed4; a.bias = bias; a.beta = beta; a.gamma = gamma; a.out = out;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.CoutP = round_up(Cout, 32); a.act = act;
  a.ticket = (unsigned long long*)ticket;
  a.s2d = 0; a.s2d_in = 0; a.nphase = 4; a.u_phase_stride = dsic_wino_weight_floats(Cout, Cin);
  return wino_launch(a, (hipStream_t)stream);
}

extern "C" int dsic_conv3x3_wino_nhwc(const float* in, const float* u_packed, const float* bias,
                                      const float* beta, const float* gamma, float* out, int B, int H,
                                      int W, int Cin, int Cout, int act, int s2d_out, int s2d_in,
                                      void* ticket, void* stream) {
  DSIC_REQUIRE(in && u_packed && bias && out && ticket, "conv3x3_wino: null pointer");
  DSIC_REQUIRE(B > 0 && H > 0 && W > 0, "conv3x3_wino: empty tensor");
  DSIC_REQUIRE(Cin > 0 && Cin % 32 == 0, "conv3x3_wino: Cin=%d must be a positive multiple of 32", Cin);
  DSIC_REQUIRE(Cout > 0 && Cout % 4 == 0 && Cout <= 128, "conv3x3_wino: Cout=%d must be a multiple of 4, <= 128", Cout);
  DSIC_REQUIRE(act >= 0 && act <= 3, "conv3x3_wino: act=%d", act);
  DSIC_REQUIRE(!(act == DSIC_ACT_GDN || act == DSIC_ACT_IGDN) || (beta && gamma), "conv3x3_wino: GDN needs beta and gamma");
  WinoArgs a{};
  a.in = in; a.u = u_packed; a.bias = bias; a.beta = beta; a.gamma = gamma; a.out = out;
  DSIC_REQUIRE(!s2d_out || (H % 2 == 0 && W % 2 == 0), "conv3x3_wino: space-to-depth output needs even H and W");
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.CoutP = round_up(Cout, 32); a.act = act;
  a.s2d = s2d_out;
  a.ticket = (unsigned long long*)ticket;
  DSIC_REQUIRE(!s2d_in || Cin % 128 == 0, "conv3x3_wino: space-to-depth input needs Cin = 4*Cs with Cs %% 32 == 0");
  a.s2d_in = s2d_in ? 1 : 0;
  a.nphase = 1; a.u_phase_stride = 0;
  return wino_launch(a, (hipStream_t)stream);
}

static int wino_launch(WinoArgs& a, hipStream_t st) {
  const int B = a.B, H = a.H, W = a.W;
  a.tiles_x = ceil_div(W, 16); a.tiles_y = ceil_div(H, 8);
  const int64_t nt = (int64_t)a.tiles_x * a.tiles_y * B * a.nphase;
  DSIC_REQUIRE(nt < ((int64_t)1 << 31), "conv3x3_wino: too many tiles");
  DSIC_REQUIRE((int64_t)H * W * a.Cin * 4 < ((int64_t)1 << 31) &&
                   (int64_t)H * W * a.Cout * 4 * (a.nphase == 4 ? 4 : 1) < ((int64_t)1 << 31),
               "conv3x3_wino: one image must stay below 2 GiB (32-bit offsets inside an image)");
  DSIC_REQUIRE((int64_t)16 * (a.Cin / 8) * a.CoutP * 8 * 4 * a.nphase < ((int64_t)1 << 31),
               "conv3x3_wino: transformed weights must stay below 2 GiB");
  a.ntiles = (int)nt;
  // Small outputs are read back by the next layer from L2/MALL (cached stores measured 1-3 % faster per
  // step); an output that cannot stay there anyway is streamed (1 % faster per layer).
  a.nt_out = (int64_t)B * H * W * a.Cout * 4 * (a.nphase == 4 ? 4 : 1) > WINO_NT_BYTES;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  static bool attr_set_dev[64] = {};
  bool& attr_set = attr_set_dev[dev];  // per device: the attribute belongs to the device's code object
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_wino_kernel<0>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, WLDS_TOTAL);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)conv_wino_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              WLDS_TOTAL);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)conv_wino_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              WLDS_TOTAL);
    if (e != hipSuccess) {
      set_error("conv3x3_wino: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return DSIC_EHIP;
    }
    attr_set = true;
  }
  // persistent: one workgroup per CU.  DSIC_WINO_GRID lowers the count when the launching stream
  // owns fewer CUs (CU-masked streams, dsic_stream_create_masked).
  // one persistent workgroup per compute unit of THIS device (DSIC_WINO_GRID overrides it for experiments)
  static int max_grid_dev[64] = {};
  if (max_grid_dev[dev] == 0) {
    const char* g = getenv("DSIC_WINO_GRID");
    int n = g ? atoi(g) : 0;
    if (n < 1 || n > 1024) {
      if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
    }
    max_grid_dev[dev] = n;
  }
  const int max_grid = max_grid_dev[dev];
  const int grid = a.ntiles < max_grid ? a.ntiles : max_grid;
  if (a.s2d_in)
    hipLaunchKernelGGL(conv_wino_kernel<1>, dim3(grid), dim3(WTHREADS), WLDS_TOTAL, st, a);
  else if (a.nphase == 4)
    hipLaunchKernelGGL(conv_wino_kernel<2>, dim3(grid), dim3(WTHREADS), WLDS_TOTAL, st, a);
  else
    hipLaunchKernelGGL(conv_wino_kernel<0>, dim3(grid), dim3(WTHREADS), WLDS_TOTAL, st, a);
  return check_launch("conv3x3_wino");
}
