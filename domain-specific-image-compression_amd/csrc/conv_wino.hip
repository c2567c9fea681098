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
// Workgroup = 512 threads = 8 waves, persistent over output tiles of 16x8 pixels
// (= 8x4 Winograd tiles = the 32 rows of one MFMA M tile).  The 16 Winograd
// positions are 16 independent GEMMs  D_p[32 tiles][Cout] += V_p[32][Cin] U_p[Cin][Cout].
// Wave w owns column tile nt = w&3 (32 output channels) and positions
// 8*(w>>2) .. +7: 8 accumulators of 32x32 = 128 VGPRs, two waves per SIMD.
// Per 32-channel chunk every thread loads 12 float4 of the NHWC input
// (prefetched one chunk ahead, across tiles), transforms its (tile, 4 channels,
// half of the positions) with 16 float4 add/subs and writes V into a
// double-buffered LDS image [pos][tile][36]; U (transformed weights, packed
// [pos][Cin/8][CoutP][8]) streams from L2 into registers.  Epilogue: the two
// position halves exchange partial inverse transforms through LDS, each wave
// finishes one output row parity, applies bias + activation and stores 16 bytes
// per lane through the LDS transpose.
#include <stdlib.h>

#include "common.h"

#ifndef WINO_STAMP
#define WINO_STAMP 0
#endif

namespace dsic {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

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
};

__device__ __forceinline__ float wino_act(float v, int act, float beta, float gamma) {
  if (act == DSIC_ACT_GDN || act == DSIC_ACT_IGDN) {
    return gdn_apply(v, beta, gamma, act == DSIC_ACT_IGDN);
  } else if (act == DSIC_ACT_RELU) {
    return v > 0.f ? v : 0.f;
  }
  return v;
}

#if WINO_STAMP
__device__ long long wino_stamps[256 * 32];
#define STAMP(i) if (WINO_STAMP && lane == 0 && wave == stamp_wave && tile_count == 2) wino_stamps[blockIdx.x * 32 + (i)] = __builtin_amdgcn_s_memtime()
#else
#define STAMP(i)
#endif

constexpr int WCK = 32;                    // channels per chunk
constexpr int WP = WCK + 4;                // LDS floats per (pos, tile) row
constexpr int WBUF = 16 * 32 * WP;         // floats per V buffer
constexpr int WLDS_BYTES = 2 * WBUF * 4;   // 147456
constexpr int WLDS_TOTAL = WLDS_BYTES + 16;  // + the next-tile mailbox

// ZSKIP: the launch has structurally zero Winograd positions (space-to-depth input or
// ConvTranspose2d phases) whose MFMA clusters are skipped; plain 3x3 layers use ZSKIP = false
// and carry no test in the loop.
// Register budget: the kernel must stay at <= 240 VGPRs.  Two of these waves per SIMD then leave 32
// of the 512-entry file, so a small wave of another stream (the serial range coder, 24 VGPRs) can
// stay resident beside this persistent kernel instead of waiting for a CU to drain.  Hence the U
// ring is only two deep (a deeper ring measured no faster).
template <bool ZSKIP>
__global__ __launch_bounds__(512, 2) void conv_wino_kernel(const WinoArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const int nt = wave & 3, ph = wave >> 2;
  const int Cin = a.Cin, Cin8 = Cin >> 3;
  const int nchunks = Cin / WCK;
  const bool nvalid = nt * 32 < a.CoutP;
  const int wstep = a.CoutP * 8;
  const unsigned boff = (unsigned)((((nvalid ? nt : 0) * 32 + l31) * 8 + 4 * h) * 4);  // bytes

  // producer role of this thread: Winograd tile pt, channel quad pq, position half pr
  const int pt = tid >> 4, pq = tid & 7, pr = (tid >> 3) & 1;  // 8 adjacent lanes = 128 contiguous bytes
  const int ptx = pt & 7, pty = pt >> 3;
  // pr = 0 computes xi in {0,1} from input rows 0,1,2; pr = 1 computes xi in {2,3} from rows 1,2,3
  const int vwrite = ((pr * 8) * 32 + pt) * WP + 4 * pq;  // + (xi_local*4 + nu)*32*WP

  floatx4 raw[12];
  int cur_tile = blockIdx.x;
  // ---- producer: the 3x4 pixel patch of (tile, chunk) this thread loads ---------------------
  // Loads are issued ONE AT A TIME between MFMA clusters (vmcnt retires in order: a burst of
  // 12 loads in front of the U-fragment loads would stall the first cluster of every chunk
  // for a full memory latency).  Out-of-image pixels load a clamped address and are zeroed
  // in the transform, so the load itself is branch-free.
  int ld_gy0 = 0, ld_gx0 = 0;
  bool ld_border = true;  // the 18x10 window of the aimed tile leaves the image somewhere
  const char* ld_base = (const char*)a.in;  // wave-uniform: image + chunk; lanes add a 32-bit byte offset
  auto aim = [&](int item, int chunk) {
    const int tile = item / a.nphase;
    const int tx = tile % a.tiles_x;
    const int ty = (tile / a.tiles_x) % a.tiles_y;
    const int n = tile / (a.tiles_x * a.tiles_y);
    ld_border = ty == 0 || tx == 0 || ty * 8 + 9 > a.H || tx * 16 + 17 > a.W;
    ld_gy0 = ty * 8 + 2 * pty - 1 + pr;
    ld_gx0 = tx * 16 + 2 * ptx - 1;
    ld_base = (const char*)(a.in + (size_t)n * a.H * a.W * Cin + chunk * WCK);
  };
  auto load_one = [&](int k) {
    int gy = ld_gy0 + (k >> 2), gx = ld_gx0 + (k & 3);
    gy = gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy);
    gx = gx < 0 ? 0 : (gx >= a.W ? a.W - 1 : gx);
    const unsigned off = (unsigned)(((gy * a.W + gx) * Cin + 4 * pq) * 4);
    raw[k] = *(const floatx4*)(ld_base + off);
  };
  // ---- producer: B^T d B for this thread's two xi rows, in 8 pieces that are issued between
  // MFMA clusters (VALU and LDS writes ride in the shadow of the 64-cycle MFMAs) ---------------
  // Columns first, in place (d B needs one window row at a time and overwrites it), rows second
  // with the results stored straight to LDS: no intermediate tile, 32 fewer live registers.
  //   pieces 0..2: local row i:  (w0,w1,w2,w3) = (d0-d2, d1+d2, d2-d1, d1-d3), out-of-image zeroed
  //   pieces 3..6: xi row x = (piece-3)>>1, nu pair (piece-3)&1:  xi0 = r0-r2, xi1 = r1+r2  (pr=0)
  //                                                               xi2 = r1-r0, xi3 = r0-r2  (pr=1, rows d1,d2,d3)
  auto transform_piece = [&](int piece, float* vbuf) {
    if (piece < 3) {
      const int i = piece;
      floatx4 d0 = raw[i * 4 + 0], d1 = raw[i * 4 + 1], d2 = raw[i * 4 + 2], d3 = raw[i * 4 + 3];
      if (ld_border) {
        const int gy = ld_gy0 + i;
        const bool yok = gy >= 0 && gy < a.H;
        const floatx4 z = {0.f, 0.f, 0.f, 0.f};
        if (!(yok && ld_gx0 >= 0 && ld_gx0 < a.W)) d0 = z;
        if (!(yok && ld_gx0 + 1 >= 0 && ld_gx0 + 1 < a.W)) d1 = z;
        if (!(yok && ld_gx0 + 2 >= 0 && ld_gx0 + 2 < a.W)) d2 = z;
        if (!(yok && ld_gx0 + 3 >= 0 && ld_gx0 + 3 < a.W)) d3 = z;
      }
      raw[i * 4 + 0] = d0 - d2;
      raw[i * 4 + 1] = d1 + d2;
      raw[i * 4 + 2] = d2 - d1;
      raw[i * 4 + 3] = d1 - d3;
    } else {
      const int x = (piece - 3) >> 1, half = (piece - 3) & 1;
      float* dst = vbuf + vwrite + (x * 4 + 2 * half) * 32 * WP;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int nu = 2 * half + q;
        floatx4 v;
        if (pr == 0)
          v = x == 0 ? raw[0 * 4 + nu] - raw[2 * 4 + nu] : raw[1 * 4 + nu] + raw[2 * 4 + nu];
        else
          v = x == 0 ? raw[1 * 4 + nu] - raw[0 * 4 + nu] : raw[0 * 4 + nu] - raw[2 * 4 + nu];
        *(floatx4*)(dst + q * 32 * WP) = v;
      }
    }
  };

  floatx16 acc[8];
  int tile_count = 0;
  const int stamp_wave = 0;
  (void)tile_count; (void)stamp_wave;
  const int aread = ((ph * 8) * 32 + l31) * WP + 4 * h;  // + p*32*WP + sub*8

  // Tiles are handed out dynamically (first round = blockIdx.x, then a global ticket): a CU that
  // is slowed down - e.g. by co-resident waves of another stream - simply takes fewer tiles,
  // instead of stretching the tail of every launch.  Arithmetic per tile is unchanged.
  volatile int* mailbox = (volatile int*)(lds + 2 * WBUF);  // [0]: tile after the current one
  if (tid == 0) mailbox[0] = (int)(atomicAdd(a.ticket, 1ULL) + gridDim.x);
  if (cur_tile < a.ntiles) {
    aim(cur_tile, 0);
#pragma unroll
    for (int k = 0; k < 12; ++k) load_one(k);
#pragma unroll
    for (int piece = 0; piece < 7; ++piece) transform_piece(piece, lds);
    __syncthreads();
  }
  int buf = 0;
  floatx4 Bq[2];
#pragma unroll
  for (int f = 0; f < 1; ++f)
    Bq[f] = *(const floatx4*)((const char*)(a.u + (size_t)((cur_tile < a.ntiles ? cur_tile : 0) % a.nphase) * a.u_phase_stride +
                                           (size_t)((ph * 8 + f) * Cin8) * wstep) + boff);
  Bq[1] = Bq[0];
  while (cur_tile < a.ntiles) {
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;
    const int next_tile = mailbox[0];  // written before the last barrier every thread has passed
    const float* u_cur = a.u + (size_t)(cur_tile % a.nphase) * a.u_phase_stride;
    const float* u_nxt = a.u + (size_t)((next_tile < a.ntiles ? next_tile : cur_tile) % a.nphase) * a.u_phase_stride;
    tile_count++;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
      STAMP(chunk * 4 + 0);
      // aim the producer at the next (tile, chunk); past the end it re-reads valid data
      const bool last = chunk + 1 == nchunks;
      const bool have_next = !last || next_tile < a.ntiles;
      aim(last ? (next_tile < a.ntiles ? next_tile : cur_tile) : cur_tile, last ? 0 : chunk + 1);
      // consumer: 8 positions x 4 sub-chunks x 4 k-steps.  U fragments run in a 4-deep
      // register ring three steps ahead (L2 latency), continuous across chunks and tiles
      // (U does not depend on the tile); V fragments one step ahead (LDS latency).
      // Structurally zero Winograd positions: a 3-tap filter with a zero end tap has a zero
      // transform component (G (g0,g1,0)^T)[3] = 0, (G (0,g1,g2)^T)[0] = 0.  For the space-to-depth
      // form of a 5x5/s2 kernel the phases a=1 / b=1 lack the last row / column (xi=3 / nu=3
      // vanish); for a ConvTranspose2d phase py=1 / px=1 lacks the first row / column (xi=0 /
      // nu=0 vanish).  Those MFMA clusters are skipped: 49 instead of 64 position-phase GEMMs.
      unsigned zero_xi = 4, zero_nu = 4, zero_xi_n = 4, zero_nu_n = 4;  // this chunk / the next one; 4 = none
      if (ZSKIP && a.s2d_in) {
        const int per = nchunks >> 2;
        const int blk = chunk / per;  // channel block (a,b) = (blk>>1, blk&1)
        const int blk_n = last ? 0 : (chunk + 1) / per;
        if (blk >> 1) zero_xi = 3;
        if (blk & 1) zero_nu = 3;
        if (blk_n >> 1) zero_xi_n = 3;
        if (blk_n & 1) zero_nu_n = 3;
      } else if (ZSKIP && a.nphase == 4) {
        const int phase = cur_tile & 3;  // grid is a multiple of 4: the next item has the same phase
        if (phase >> 1) zero_xi = zero_xi_n = 0;
        if (phase & 1) zero_nu = zero_nu_n = 0;
      }
      auto is_zero = [&](int f) {  // step f of this chunk (f < 32) or f-32 of the next
        const unsigned xi = (unsigned)(ph * 2 + ((f & 7) >> 2)), nu = (unsigned)(f & 3);
        return f < 32 ? (xi == zero_xi || nu == zero_nu) : (xi == zero_xi_n || nu == zero_nu_n);
      };
      const float* vb = lds + buf * WBUF + aread;
      const float* ub = u_cur + (size_t)(chunk * 4) * wstep;
      const float* ubn = last ? u_nxt : u_cur + (size_t)((chunk + 1) * 4) * wstep;
      floatx4 Aq[2];
      Aq[0] = *(const floatx4*)(vb);
#pragma unroll
      for (int it = 0; it < 32; ++it) {  // it = sub*8 + p
        {
          const int f = it + 1;  // U fragment to fetch now
          const int fp = f & 7, fs = (f >> 3) & 3;
          const float* src = (f < 32 ? ub : ubn) + (size_t)((ph * 8 + fp) * Cin8 + fs) * wstep;  // uniform
          Bq[f & 1] = *(const floatx4*)((const char*)src + boff);
        }
        if (it < 12) load_one(it);  // producer loads first: they are consumed from step 20 on
        if (it + 1 < 32) {
          const int p1 = (it + 1) & 7, s1 = (it + 1) >> 3;
          Aq[(it + 1) & 1] = *(const floatx4*)(vb + p1 * 32 * WP + s1 * 8);
        }
        const int p0 = it & 7;
        if (!ZSKIP || !is_zero(it)) {  // wave-uniform
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int s = 0; s < 4; ++s)
            acc[p0] = __builtin_amdgcn_mfma_f32_32x32x2f32(Aq[it & 1][s], Bq[it & 1][s], acc[p0], 0, 0, 0);
          __builtin_amdgcn_s_setprio(0);
        }
        if (it >= 20 && it < 27 && have_next) transform_piece(it - 20, lds + (buf ^ 1) * WBUF);
      }
      STAMP(chunk * 4 + 1);
      STAMP(chunk * 4 + 2);
      __syncthreads();
      STAMP(chunk * 4 + 3);
      buf ^= 1;
    }

    // ---- inverse transform + epilogue ------------------------------------------------------
    // Wave (nt, ph) holds M[xi][nu] for xi in {2ph, 2ph+1}.  N[xi][j] = (M A)[xi][j]:
    //   N[.][0] = M0 + M1 + M2,  N[.][1] = M1 - M2 - M3.
    // Y[i][j] = (A^T N)[i][j]:  Y[0] = N0 + N1 + N2,  Y[1] = N1 - N2 - N3.
    // ph=0 finishes row i=0 and needs N2 from ph=1; ph=1 finishes i=1 and needs N1 from ph=0.
    STAMP(24);
    // Register-lean form: keep only this wave's own partial sum per j and hand the other term
    // to the partner straight away.
    //   ph=0: own = N0 + N1, sends N1;   ph=1: own = -(N2 + N3), sends N2;   Y = own + received.
    float* xch = lds + (buf ^ 1) * WBUF;  // 4 nt x 2 ph x 2 j x 16 e x 64 lanes = 16384 floats
    floatx16 own[2];
    {
      float* dst = xch + ((nt * 2 + ph) * 2) * 1024 + lane;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const floatx16 na = j == 0 ? (acc[0] + acc[1]) + acc[2] : (acc[1] - acc[2]) - acc[3];  // local xi 0
        const floatx16 nb = j == 0 ? (acc[4] + acc[5]) + acc[6] : (acc[5] - acc[6]) - acc[7];  // local xi 1
        own[j] = ph == 0 ? na + nb : -(na + nb);
#pragma unroll
        for (int e = 0; e < 16; ++e) dst[j * 1024 + e * 64] = ph == 0 ? nb[e] : na[e];
      }
    }
    __syncthreads();
    floatx16 yv[2];
    {
      const float* src = xch + ((nt * 2 + (ph ^ 1)) * 2) * 1024 + lane;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) yv[j][e] = own[j][e] + src[j * 1024 + e * 64];
    }
    __syncthreads();  // exchange area free again (it is the next chunk's transform target)
    STAMP(25);

    if (nvalid) {
      const int stile = cur_tile / a.nphase, phase = cur_tile % a.nphase;
      const int tx = stile % a.tiles_x;
      const int ty = (stile / a.tiles_x) % a.tiles_y;
      const int n = stile / (a.tiles_x * a.tiles_y);
      const int up = a.nphase == 4 ? 2 : 1, ppy = phase >> 1, ppx = phase & 1;  // sub-pixel phase placement
      const int col = nt * 32 + l31;
      const bool cok = col < a.Cout;
      const float bias = cok ? a.bias[col] : 0.f;
      float beta = 1.f, gamma = 0.f;
      if (cok && (a.act == DSIC_ACT_GDN || a.act == DSIC_ACT_IGDN)) {
        beta = a.beta[col];
        gamma = a.gamma[col];
      }
      // per-wave transpose patch: overlays the exchange area, which is free after the barrier above
      float* epi = xch + wave * (32 * 36);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int rr = (e & 3) + 8 * (e >> 2) + 4 * h;
          epi[rr * 36 + l31] = wino_act(__fadd_rn(yv[j][e], bias), a.act, beta, gamma);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        const int c4 = (lane & 7) * 4;
        const int nn = nt * 32 + c4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int rr = (lane >> 3) + 8 * i;  // Winograd tile index 0..31
          const floatx4 v = *(const floatx4*)(epi + rr * 36 + c4);
          const int oy = ty * 8 + 2 * (rr >> 3) + ph;
          const int ox = tx * 16 + 2 * (rr & 7) + j;
          if (oy < a.H && ox < a.W && nn < a.Cout) {
            const size_t o = up == 2 ? (((size_t)n * (2 * a.H) + (2 * oy + ppy)) * (2 * a.W) + (2 * ox + ppx)) * a.Cout + nn
                             : a.s2d ? (((size_t)n * (a.H >> 1) + (oy >> 1)) * (a.W >> 1) + (ox >> 1)) * (4 * a.Cout) +
                                         ((oy & 1) * 2 + (ox & 1)) * a.Cout + nn
                                   : (((size_t)n * a.H + oy) * a.W + ox) * a.Cout + nn;
            *(floatx4*)(a.out + o) = v;
          }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
      }
    }
    STAMP(26);
    __syncthreads();  // transpose patches done before the next tile's producers reuse the area
    STAMP(27);
    // every thread has read mailbox[0] (next_tile) long before this point (>= 3 barriers ago)
    if (tid == 0 && next_tile < a.ntiles) mailbox[0] = (int)(atomicAdd(a.ticket, 1ULL) + gridDim.x);
    __syncthreads();
    cur_tile = next_tile;
  }
  if (tid == 0) {  // last workgroup out re-arms the ticket for the next launch on this stream
    const unsigned long long done = atomicAdd(a.ticket + 1, 1ULL);
    if (done == (unsigned long long)gridDim.x - 1) {
      a.ticket[0] = 0ULL;
      a.ticket[1] = 0ULL;
    }
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
  a.ntiles = (int)nt;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_wino_kernel<false>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, WLDS_TOTAL);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)conv_wino_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              WLDS_TOTAL);
    if (e != hipSuccess) {
      set_error("conv3x3_wino: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return DSIC_EHIP;
    }
    attr_set = true;
  }
  // persistent: one workgroup per CU.  DSIC_WINO_GRID lowers the count when the launching stream
  // owns fewer CUs (CU-masked streams, dsic_stream_create_masked).
  static int max_grid = 0;
  if (max_grid == 0) {
    const char* g = getenv("DSIC_WINO_GRID");
    max_grid = g ? atoi(g) : 256;
    if (max_grid < 1 || max_grid > 1024) max_grid = 256;
  }
  const int grid = a.ntiles < max_grid ? a.ntiles : max_grid;
  if (a.s2d_in || a.nphase == 4)
    hipLaunchKernelGGL(conv_wino_kernel<true>, dim3(grid), dim3(512), WLDS_TOTAL, st, a);
  else
    hipLaunchKernelGGL(conv_wino_kernel<false>, dim3(grid), dim3(512), WLDS_TOTAL, st, a);
  return check_launch("conv3x3_wino");
}
