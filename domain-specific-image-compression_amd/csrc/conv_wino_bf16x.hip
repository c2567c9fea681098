// Winograd F(2x2,3x3), split-bf16 MFMA - the kernel of conv_wino_bf16.hip with a different wave budget.
//
// conv_wino_bf16.hip runs 8 MFMA waves + 4 helper waves, three waves per SIMD, 168 VGPRs each.  Its MFMA waves wait
// for their transformed-weight (U) fragments most of the time: 128 accumulator registers leave room for two
// fragment pairs in flight, requested one position step ahead, and L2 answers later than that (DESIGN.md section 3:
// a 4-deep ring is worth 13 % and does not fit).  Here a workgroup is EIGHT waves, two per SIMD, 256 VGPRs each,
// and every wave does both jobs of a 16-channel chunk period:
//   M(c)  its 8 position steps of chunk c (24 MFMAs) on V[c&1], U fragments from a ring RING steps deep;
//   T(c)  1/8 of the helper work: stage the input window of chunk c+2, transform + split chunk c+1 into
//         V[(c+1)&1], issue the window loads of chunk c+4, copy out a share of the previous tile's outputs.
// M(c) and T(c) touch different buffers, so their order inside a period is free: the waves of position half 0 run
// M then T, those of half 1 T then M.  A SIMD hosts one wave of each half - while one feeds the matrix pipe the
// other uses the VALU and LDS.  One barrier per period, as before.
// Work split of T: wave w transforms the 4 positions (xi in {2hx, 2hx+1}) x (nu in {2hn, 2hn+1}), hx = (w>>1)&1,
// hn = w&1, of the Winograd tiles 16(w>>2) .. +15 (lane = tile x channel quad): three patch rows x three columns.
// Same LDS layout, position tables, fold and epilogue as conv_wino_bf16.hip (conv_wino_bf16.h); no split-K.
#include "conv_wino_bf16.h"

#ifndef WBX_RING
#define WBX_RING 4
#endif
#ifndef WBX_ADOUBLE
#define WBX_ADOUBLE 1
#endif

namespace dsic {
namespace wbx {

using namespace dsic::wb;

constexpr int XTHREADS = 512;
constexpr int XRING = WBX_RING;  // position steps of U fragments in flight per wave; must divide 8
static_assert(8 % XRING == 0, "ring must divide the 8 steps of a chunk");

template <int MODE, bool NT_OUT>
__global__ __launch_bounds__(XTHREADS) void conv_wino_bf16x_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  float* const yreg = (float*)(lds_raw + YOFF);
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
  const int nchunks = a.Cin / CK;  // even, >= 4 (host)
  const int Cin = a.Cin;

  // ------------------------------------ transform role -------------------------------------------
  const int hx = (wave >> 1) & 1, hn = wave & 1;
  const int pt = (wave >> 2) * 16 + (lane >> 2), pq = lane & 3;
  const int ptx = pt & 7, pty = pt >> 3;
  const int vwrite = pt * ROWB + ((((pq >> 1) ^ ((pt >> 3) & 1))) << 4) + ((pq & 1) << 3);
  unsigned char* const vmine = lds_raw + vwrite + ((2 * hx) * 4 + 2 * hn) * POSB;  // + buffer + plane + local pos
  // copy-out: thread = (Winograd tile ot, channel quad oq of a 32-channel group, two of the four outputs ij)
  const int ot = (tid & 255) >> 3, oq = tid & 7, ijh = __builtin_amdgcn_readfirstlane(tid >> 8);
  const int otx = ot & 7, oty = ot >> 3;
  const int yread = ot * WP + 4 * oq;
  auto post = [&](int s, int item) {  // thread 0 only
    const int tile = item >> pshift;
    const int row = tile / a.tiles_x;
    const intx4 v = {item, tile - row * a.tiles_x, row % a.tiles_y, row / a.tiles_y};
    *(intx4*)(slots + 4 * s) = v;
  };
  // input window: 720 float4 items for 512 threads (two per thread, the second only for tid < 208)
  struct WinAim {
    unsigned off[2];
    __amdgpu_buffer_rsrc_t rsrc;
  };
  WinAim am;
  unsigned stage_off[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int i = tid + XTHREADS * j;
    stage_off[j] = (unsigned)((i >> 2) * (CK * 4) + (i & 3) * 16);
  }
  auto aim = [&](WinAim& m, const Tile& t) {
    m.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (size_t)t.n * a.H * a.W * Cin), 0, a.H * a.W * Cin * 4,
                                               0x00020000);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int i = tid + XTHREADS * j;
      const int pix = i >> 2, q = i & 3;
      const int wy = pix / WINW, wx = pix - wy * WINW;
      const int gy = t.ty * 8 - 1 + wy, gx = t.tx * 16 - 1 + wx;
      const bool ok = i < WINW * WINH * 4 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      m.off[j] = ok ? (unsigned)(((gy * a.W + gx) * Cin + 4 * q) * 4) : 0x80000000u;
    }
  };
  auto aim_nowhere = [&](WinAim& m) {
#pragma unroll
    for (int j = 0; j < 2; ++j) m.off[j] = 0x80000000u;
  };
  auto issue = [&](floatx4 (&r)[2], int chunk) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
      r[j] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(am.rsrc, am.off[j], chunk * (CK * 4), 0));
  };
  auto stage = [&](const floatx4 (&r)[2], int wbuf) {
    unsigned char* wb = lds_raw + STAGEOFF + wbuf * WINB;
    *(floatx4*)(wb + stage_off[0]) = r[0];
    if (tid < WINW * WINH * 4 - XTHREADS) *(floatx4*)(wb + stage_off[1]) = r[1];
  };
  // this thread's patch inside a staged window: rows 2*pty+hx .. +2, columns 2*ptx+hn .. +2, quad pq
  const int patch0 = ((2 * pty + hx) * WINW + 2 * ptx + hn) * (CK * 4) + pq * 16;
  auto split_store = [&](floatx4 v, unsigned char* dst) {
    const unsigned h0 = cvt_pk_bf16(v[0], v[1]), h1 = cvt_pk_bf16(v[2], v[3]);
    *(uintx2*)dst = uintx2{h0, h1};
    const float r0 = v[0] - bf16_lo(h0), r1 = v[1] - bf16_hi(h0), r2 = v[2] - bf16_lo(h1), r3 = v[3] - bf16_hi(h1);
    const unsigned m0 = cvt_pk_bf16(r0, r1), m1 = cvt_pk_bf16(r2, r3);
    *(uintx2*)(dst + PLANEB) = uintx2{m0, m1};
  };
  // B^T d B for this thread's 2 x 2 positions, in slices that the period interleaves with its MFMA steps:
  //   t_load(wbuf)            three patch rows x three columns from the staged window, row transform -> xq[r][k]
  //   t_pos(j, vb, zxi, znu)  position (r = j>>1, s = j&1): column transform, split, store into V buffer vb
  // zxi / znu: the structurally zero row / column of the chunk (4 = none)
  floatx4 xq[2][3];
  auto t_load = [&](int wbuf) {
    const unsigned char* src = lds_raw + STAGEOFF + wbuf * WINB + patch0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const floatx4 ra = *(const floatx4*)(src + (0 * WINW + k) * (CK * 4));
      const floatx4 rb = *(const floatx4*)(src + (1 * WINW + k) * (CK * 4));
      const floatx4 rc = *(const floatx4*)(src + (2 * WINW + k) * (CK * 4));
      if (hx == 0) {   // wave-uniform
        xq[0][k] = ra - rc;   // xi 0: r0 - r2
        xq[1][k] = rb + rc;   // xi 1: r1 + r2
      } else {
        xq[0][k] = rb - ra;   // xi 2: r2 - r1
        xq[1][k] = ra - rc;   // xi 3: r1 - r3
      }
    }
  };
  // columns (x0..x3 = patch columns): nu 0: x0-x2, 1: x1+x2 (hn 0: local 0,1,2 = x0,x1,x2);
  //                                   nu 2: x2-x1, 3: x1-x3 (hn 1: local 0,1,2 = x1,x2,x3)
  auto t_pos = [&](int j, int vb, unsigned zxi, unsigned znu) {
    const int r = j >> 1, sc = j & 1;
    const bool live = zxi != (unsigned)(2 * hx + r) && znu != (unsigned)(2 * hn + sc);   // wave-uniform
    if (!live) return;
    const floatx4 v = hn == 0 ? (sc == 0 ? xq[r][0] - xq[r][2] : xq[r][1] + xq[r][2])
                              : (sc == 0 ? xq[r][1] - xq[r][0] : xq[r][0] - xq[r][2]);
    split_store(v, vmine + vb * VBUFB + (r * 4 + sc) * POSB);
  };
  auto commit = [&](int wbuf, int vb, unsigned zxi, unsigned znu) {
    t_load(wbuf);
#pragma unroll
    for (int j = 0; j < 4; ++j) t_pos(j, vb, zxi, znu);
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
  struct OutAim {
    unsigned po[2];
    __amdgpu_buffer_rsrc_t rs;
  };
  auto next_ticket = [&]() { return (int)(atomicAdd(a.ticket, 1ULL) + gridDim.x); };
  auto aim_out = [&](OutAim& o, const Tile& t) {
    const int phase = t.item & (a.nphase - 1);
    const int ppy = phase >> 1, ppx = phase & 1;
    const int OH = a.nphase == 4 ? 2 * a.H : a.H, OW = a.nphase == 4 ? 2 * a.W : a.W;
    o.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)t.n * OH * OW * a.ostride + a.ooff), 0,
                                             (OH * OW * a.ostride - a.ooff) * 4, 0x00020000);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int ij = 2 * ijh + k;
      const int oy = t.ty * 8 + 2 * oty + (ij >> 1);
      const int ox = t.tx * 16 + 2 * otx + (ij & 1);
      const unsigned po =
          (unsigned)(a.nphase == 4 ? ((2 * oy + ppy) * OW + (2 * ox + ppx)) * a.ostride
                     : a.s2d       ? ((oy >> 1) * (a.W >> 1) + (ox >> 1)) * (4 * a.Cout) + ((oy & 1) * 2 + (ox & 1)) * a.Cout
                                   : (oy * a.W + ox) * a.ostride) * 4u + 16u * oq;
      o.po[k] = oy < a.H && ox < a.W ? po : 0x80000000u;
    }
  };
  // one 32-channel group of the finished outputs: 2 LDS reads + 2 buffer stores per thread, no branch (a store
  // beyond Cout or the image gets the out-of-range offset the hardware drops)
  auto store_group = [&](const OutAim& o, int g) {
    const float* src = yreg + yread;
    const bool chan_ok = g * 32 + 4 * oq < a.Cout;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int ij = 2 * ijh + k;
      const floatx4 v = *(const floatx4*)(src + (ij * 4 + (g & 3)) * 32 * WP);
      const unsigned off = chan_ok ? o.po[k] : 0x80000000u;
      if (NT_OUT)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, v), o.rs, off, g * 128, 2);
      else
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, v), o.rs, off, g * 128, 0);
    }
  };

  // -------------------------------------- MFMA role -----------------------------------------------
  const int h = lane >> 5, l31 = lane & 31;
  const int nt = wave & 3, ph = wave >> 2;
  const bool nvalid = nt * 32 < a.CoutP;
  constexpr bool ZSKIP = MODE != 0;
  const unsigned plane_b = (unsigned)a.CoutP * 32u;
  const unsigned chunk_b = plane_b * (unsigned)P;
  const unsigned pos_b = chunk_b * (unsigned)nchunks;
  const unsigned ulane = (unsigned)((((nvalid ? nt : 0) * 32 + l31) * 2 + h) * 16);
  const int aread = l31 * ROWB + ((h ^ ((l31 >> 3) & 1)) << 4);
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
  const __amdgpu_buffer_rsrc_t urs =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.u, 0, (int)(16u * pos_b * (unsigned)a.nphase), 0x00020000);

  // ------------------------------------- prologue --------------------------------------------------
  if (tid == 0) {
    post(0, (int)blockIdx.x);
    post(1, next_ticket());
  }
  __syncthreads();  // P0
  Tile cur = read_slot(0);
  floatx4 R0[2], R1[2];
  int ticket_pre = a.ntiles;
  aim(am, cur);
  issue(R0, 0);
  issue(R1, 1);
  stage(R0, 0);
  stage(R1, 1);
  issue(R0, 2);
  issue(R1, 3);
  if (tid == 0 && read_slot(1).item < a.ntiles) ticket_pre = next_ticket();
  __syncthreads();  // P1: windows of chunks 0 and 1 are staged
  {
    unsigned zxi, znu;
    zero_of(cur, 0, zxi, znu);
    commit(0, 0, zxi, znu);    // V[0] = (cur, 0)
  }
  __syncthreads();  // P

  auto body = [&](auto ph_tag) {
    constexpr int PH = decltype(ph_tag)::value;
    constexpr unsigned ptab = PosTab<MODE, PH>::value;
    auto pos_of = [](int pi) { return (int)((ptab >> (4 * pi)) & 15u); };   // global position xi*4 + nu of step pi
    floatx16 acc[8];
    bf16x8 Bq[XRING][P];
    auto soff_item = [&](const Tile& t) { return (unsigned)(t.item & (a.nphase - 1)) * (unsigned)a.u_phase_bytes; };
    unsigned soff_phase = soff_item(cur);
    auto soff_of = [&](unsigned phase_off, int chunk, int pi) {
      return phase_off + (unsigned)pos_of(pi) * pos_b + (unsigned)chunk * chunk_b;
    };
    auto fetch = [&](bf16x8 (&dst)[P], unsigned so) {
#pragma unroll
      for (int q = 0; q < P; ++q)
        dst[q] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(urs, ulane, so + (unsigned)q * plane_b, 0));
    };
#pragma unroll
    for (int f = 0; f < XRING; ++f) fetch(Bq[f], soff_of(soff_phase, 0, f));
    int s_nxt = 1, s_wr = 2;
    OutAim oa;
    oa.rs = am.rsrc;
    oa.po[0] = oa.po[1] = 0x80000000u;
    int tile_count = 0;
    (void)tile_count;
    while (cur.item < a.ntiles) {
      tile_count++;
      const Tile nxt = read_slot(s_nxt);
      const bool more = nxt.item < a.ntiles;
      const unsigned soff_phase_nxt = soff_item(more ? nxt : cur);
      // ---- M(c): the 8 position steps of chunk c ----
      auto mfma_chunk = [&](auto first_tag, floatx4 (&R)[2], int chunk) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const bool last = chunk + 1 == nchunks;
        // T slices of this period: window of chunk c+2 staged, chunk c+1 transformed into V[(c+1)&1]
        unsigned tzxi, tznu;
        if (chunk + 1 < nchunks) zero_of(cur, chunk + 1, tzxi, tznu); else zero_of(nxt, 0, tzxi, tznu);
        stage(R, chunk & 1);
        t_load((chunk + 1) & 1);
        unsigned zero_xi = 4, zero_nu = 4;
        if (MODE == 1) {
          const int blk = chunk / (nchunks >> 2);
          if (blk >> 1) zero_xi = 3;
          if (blk & 1) zero_nu = 3;
        } else if (MODE == 2) {
          const int phase = cur.item & 3;
          if (phase >> 1) zero_xi = 0;
          if (phase & 1) zero_nu = 0;
        }
        auto is_zero = [&](int pi) {
          const int p = pos_of(pi);
          const unsigned xi = (unsigned)(p >> 2), nu = (unsigned)(p & 3);
          return xi == zero_xi || nu == zero_nu;
        };
        const unsigned char* vb = lds_raw + (chunk & 1) * VBUFB + aread;
        bf16x8 Aq[WBX_ADOUBLE + 1][P];   // V fragments (WBX_ADOUBLE: those of step pi+1 are read before the MFMAs of step pi)
#pragma unroll
        for (int q = 0; q < P; ++q) Aq[0][q] = *(const bf16x8*)(vb + pos_of(0) * POSB + q * PLANEB);
#pragma unroll
        for (int pi = 0; pi < 8; ++pi) {
          const bool live = !ZSKIP || !is_zero(pi);  // wave-uniform
          if (WBX_ADOUBLE && pi + 1 < 8) {
#pragma unroll
            for (int q = 0; q < P; ++q) Aq[(pi + 1) & WBX_ADOUBLE][q] = *(const bf16x8*)(vb + pos_of(pi + 1) * POSB + q * PLANEB);
          }
          if (live) {
            const floatx16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            floatx16 c = FIRST ? zero : acc[pi];
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Aq[pi & WBX_ADOUBLE][0], Bq[pi % XRING][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Aq[pi & WBX_ADOUBLE][1], Bq[pi % XRING][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Aq[pi & WBX_ADOUBLE][0], Bq[pi % XRING][0], c, 0, 0, 0);
            acc[pi] = c;
            if (!WBX_ADOUBLE && pi + 1 < 8) {
#pragma unroll
              for (int q = 0; q < P; ++q) Aq[0][q] = *(const bf16x8*)(vb + pos_of(pi + 1) * POSB + q * PLANEB);
            }
          } else if (FIRST) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[pi][e] = 0.f;
          }
          {
            // refill this ring slot with the fragments of step pi + XRING (next chunk / next tile past 7)
            const int f = pi + XRING;
            const unsigned so = f < 8 ? soff_of(soff_phase, chunk, f)
                                      : (last ? soff_of(soff_phase_nxt, 0, f - 8) : soff_of(soff_phase, chunk + 1, f - 8));
            // a structurally zero position of THIS chunk is never multiplied and never fetched; steps of the next
            // chunk / tile (f >= 8) are fetched unconditionally
            if (!(ZSKIP && f < 8 && is_zero(f))) fetch(Bq[pi % XRING], so);
          }
          // one slice of the transform behind every MFMA step: the matrix pipe works while the VALU splits
          if (pi >= 1 && pi <= 4) t_pos(pi - 1, (chunk + 1) & 1, tzxi, tznu);
          if (pi == 5) {
            const int k4 = chunk + 4;
            if (k4 == nchunks) {   // from here on every window load is for the next tile
              if (more) aim(am, nxt); else aim_nowhere(am);
            }
            issue(R, k4 < nchunks ? k4 : k4 - nchunks);
          }
        }
      };
      auto period = [&](auto first_tag, floatx4 (&R)[2], int c) {
        WSTAMPC(PH * 4, 3 * (c - WB_STAMP_C0), 60);
        mfma_chunk(first_tag, R, c);
        WSTAMPC(PH * 4, 3 * (c - WB_STAMP_C0) + 2, 60);
        __syncthreads();  // B_c
      };
      if (tid == 0 && more) post(s_wr, ticket_pre);
      // outputs of the previous tile (the output region is rewritten at this tile's fold): one 32-channel group
      // per chunk pair in the first eight periods (both groups of a pair at once when there are only four)
      const int npair_out = nchunks >= 8 ? 4 : 2;
      store_group(oa, 0);
      if (nchunks < 8) store_group(oa, 1);
      period(std::true_type{}, R0, 0);
      period(std::false_type{}, R1, 1);
      for (int c = 2; c < 2 * npair_out; c += 2) {
        store_group(oa, nchunks >= 8 ? (c >> 1) : c);
        if (nchunks < 8) store_group(oa, c + 1);
        period(std::false_type{}, R0, c);
        period(std::false_type{}, R1, c + 1);
      }
      for (int c = 2 * npair_out; c < nchunks; c += 2) {
        period(std::false_type{}, R0, c);
        period(std::false_type{}, R1, c + 1);
      }
      soff_phase = soff_phase_nxt;
      if (tid == 0 && more && read_slot(s_wr).item < a.ntiles) ticket_pre = next_ticket();
      aim_out(oa, cur);

      WSTAMP(PH * 4, 60);
      // ---- inverse transform, bias, activation (as conv_wino_bf16.hip) ----
      float* yown = yreg + ((4 * (2 * PH) + nt) * 32 + 4 * h) * WP + l31;
      float* yoth = yreg + ((4 * (2 * (PH ^ 1)) + nt) * 32 + 4 * h) * WP + l31;
      floatx16 send[2];
      {
        const floatx16 own = fold_partial<MODE, PH, PH, 0>(acc);
#pragma unroll
        for (int e = 0; e < 16; ++e) yown[((e & 3) + 8 * (e >> 2)) * WP] = own[e];
      }
      {
        const floatx16 own = fold_partial<MODE, PH, PH, 1>(acc);
#pragma unroll
        for (int e = 0; e < 16; ++e) yown[(4 * 32 + (e & 3) + 8 * (e >> 2)) * WP] = own[e];
      }
      send[0] = fold_partial<MODE, PH, PH ^ 1, 0>(acc);
      send[1] = fold_partial<MODE, PH, PH ^ 1, 1>(acc);
      WSTAMP(PH * 4, 61);
      __syncthreads();  // E1
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
      WSTAMP(PH * 4, 62);
      __syncthreads();  // E2
      WSTAMP(PH * 4, 63);
      cur = nxt;
      const int s_old = s_nxt;
      s_nxt = s_wr;
      s_wr = s_old == 0 ? 2 : s_old - 1;
    }
    for (int g = 0; g < 4; ++g) store_group(oa, g);   // outputs of the last tile (dropped offsets if there was none)
#if WB_STAMP
    if (wave == PH * 4) {
      wb_stamps[blockIdx.x * 128 + PH * 64 + lane] = ((long long*)(lds_raw + STAMPOFF))[PH * 64 + lane];
    }
#endif
  };
  if (ph == 0)
    body(std::integral_constant<int, 0>{});
  else
    body(std::integral_constant<int, 1>{});
  if (tid == 0) {
    const unsigned long long done = atomicAdd(a.ticket + 1, 1ULL);
    if (done == (unsigned long long)gridDim.x - 1) {
      a.ticket[0] = 0ULL;
      a.ticket[1] = 0ULL;
    }
  }
}

}  // namespace wbx
}  // namespace dsic

using namespace dsic;

#if WB_STAMP
extern "C" int dsic_debug_wbx_stamps(long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(wb::wb_stamps), sizeof(long long) * 256 * 128) == hipSuccess ? 0 : 2;
}
#endif

// launcher of the fused-role kernel (called from wb_launch in conv_wino_bf16.hip when DSIC_WINO_FUSED is set)
int dsic_wbx_launch(const wb::Args& a, int grid, hipStream_t st) {
  static bool attr_set[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  if (dev < 0 || dev >= 64) dev = 0;
  if (!attr_set[dev]) {
    const void* fns[6] = {(const void*)wbx::conv_wino_bf16x_kernel<0, false>, (const void*)wbx::conv_wino_bf16x_kernel<1, false>,
                          (const void*)wbx::conv_wino_bf16x_kernel<2, false>, (const void*)wbx::conv_wino_bf16x_kernel<0, true>,
                          (const void*)wbx::conv_wino_bf16x_kernel<1, true>,  (const void*)wbx::conv_wino_bf16x_kernel<2, true>};
    for (int i = 0; i < 6; ++i) {
      const hipError_t e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, wb::LDS_TOTAL);
      if (e != hipSuccess) {
        set_error("conv_wino_bf16x: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return DSIC_EHIP;
      }
    }
    attr_set[dev] = true;
  }
#define WBX_LAUNCH(M)                                                                                                   \
  do {                                                                                                                  \
    if (a.nt_out)                                                                                                       \
      hipLaunchKernelGGL((wbx::conv_wino_bf16x_kernel<M, true>), dim3(grid), dim3(wbx::XTHREADS), wb::LDS_TOTAL, st, a);  \
    else                                                                                                                \
      hipLaunchKernelGGL((wbx::conv_wino_bf16x_kernel<M, false>), dim3(grid), dim3(wbx::XTHREADS), wb::LDS_TOTAL, st, a); \
  } while (0)
  if (a.s2d_in)
    WBX_LAUNCH(1);
  else if (a.nphase == 4)
    WBX_LAUNCH(2);
  else
    WBX_LAUNCH(0);
#undef WBX_LAUNCH
  return check_launch("conv_wino_bf16x");
}
