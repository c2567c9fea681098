// Winograd F(2x2,3x3) convolution with the 16 position GEMMs on the bf16 MFMA path and fp32-class
// results: every fp32 operand is split into bf16 planes, x = hi + mid (+ lo), and the products
// hi*hi, hi*mid, mid*hi (PLANES = 2; + mid*mid, hi*lo, lo*hi for PLANES = 3) are accumulated in
// fp32 by v_mfma_f32_32x32x16_bf16.  A bf16 x bf16 product is exact in fp32, so the only change
// against the fp32-MFMA kernel (conv_wino.hip) is the dropped low-order cross terms: relative
// 2^-16 per product for two planes, 2^-24 for three.  tools/split_bf16_emulation.py runs both on
// the 9 reference fixtures: two planes <= 1 latent flip per image, |dbpp| <= 1e-5; three planes
// 0 flips, |dbpp| <= 1e-8.  Same layers as conv_wino.hip: conv(.,.,3,1),
// conv(C,C,5,2) over space-to-depth, ConvTranspose2d(.,.,5,2,2,1) as four phases
// (code/modelv2/layers.py:54-72, 83-97, 108-124); same fused bias + GDN/IGDN/ReLU epilogue.
//
// The bf16 MFMA runs at 16x the fp32-input rate, so three of them per k-step are 5.3x less matrix
// time; what the workgroup must now feed is what shapes the kernel:
//   * chunks of 16 input channels (one MFMA k-step); V planes [plane][pos 16][tile 32][16 bf16],
//     32-byte rows with the 16-byte k-block XOR-swizzled by (tile>>3)&1 (conflict-free
//     ds_read_b128), double buffered: 64 KB.  The outputs get their own LDS region (72 KB), so the
//     V buffers never wait for the copy-out.
//   * 768 threads = 8 MFMA waves + 4 helper waves, as in conv_wino.hip.  MFMA wave (nt, ph) owns
//     32 output channels x 8 positions (128 accumulator VGPRs); per chunk and position it reads
//     the two V planes (2 ds_read_b128), takes the two U planes from its register ring
//     (transformed weights, bf16, [pos][chunk][plane][CoutP][16], streamed from L2) and issues
//     3 MFMAs.
//   * the four helper waves (one per SIMD) share every chunk.  The tile's 18x10-pixel input window is
//     loaded once (3 float4 per thread, two phases ahead) and staged in LDS (2 x 11.5 KB); thread =
//     (Winograd tile, channel quad, half of the positions) then reads three rows of its 4x4 patch
//     from there, applies B^T d B in fp32, splits into the bf16 planes and stores them into the V
//     buffer the MFMA waves are not reading.
// Tile hand-out, the fold through LDS and the fused epilogue follow conv_wino.hip.
#include "conv_wino_bf16.h"

namespace dsic {
namespace wb {

// SPLITK is a template parameter: the unsplit kernels keep their register allocation (with the work item's chunk
// offset as run-time state the 3x3 kernel spilled 20 more SGPRs into its loops and ran 20 % slower).
template <int MODE, bool NT_OUT, bool SPLITK>
__global__ __launch_bounds__(THREADS) void conv_wino_bf16_kernel(const Args a) {
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
    t.ty = SPLITK ? __builtin_amdgcn_readfirstlane(v[2]) & 0xFFFF : __builtin_amdgcn_readfirstlane(v[2]);
    t.ks = SPLITK ? __builtin_amdgcn_readfirstlane(v[2]) >> 16 : 0;
    t.n = __builtin_amdgcn_readfirstlane(v[3]);
    return t;
  };
  const int pshift = a.nphase == 4 ? 2 : 0;
  const int tchunks = a.Cin / CK;                    // = nchunks * ksplit
  const int nchunks = SPLITK ? a.kchunks : tchunks;  // chunks of one work item: even, >= 4 (host)
  const int ksplit = SPLITK ? a.ksplit : 1;

  if (wave >= 8) {
    // =================================== helper waves ===========================================
    // Every chunk is produced by all four helper waves (one per SIMD, so the VALU work is spread
    // evenly beside the MFMA waves): thread = (Winograd tile pt, channel quad pq of the 16-channel
    // chunk, row half hx).  Half hx owns the positions xi in {2hx, 2hx+1}: they need only three of
    // the four patch rows (xi 0: r0-r2, 1: r1+r2 | 2: r2-r1, 3: r1-r3).
    // the helpers' VALU stream fits beside the bf16 MFMAs (an MFMA holds the SIMD's vector issue for
    // 8 of its 32 cycles); they are the longer pole of a phase, so they win the issue arbitration
    __builtin_amdgcn_s_setprio(WB_HPRIO);
    const int ht = tid - 512;
    const int hx = __builtin_amdgcn_readfirstlane(ht >> 7);
    const int Cin = a.Cin;
    const int t7 = ht & 127;
    const int pt = t7 >> 2, pq = t7 & 3;
    const int ptx = pt & 7, pty = pt >> 3;
    // LDS byte offset of this thread's 8 bytes inside a (plane, pos) block
    const int vwrite = pt * ROWB + ((((pq >> 1) ^ ((pt >> 3) & 1))) << 4) + ((pq & 1) << 3);
    unsigned char* const vmine = lds_raw + vwrite + (2 * hx) * 4 * POSB;  // + buffer + plane + local pos
    // output side: thread = (Winograd tile ot, channel quad oq of a 32-channel group), all 256 helpers
    const int ot = ht >> 3, oq = ht & 7;
    const int otx = ot & 7, oty = ot >> 3;
    const int yread = ot * WP + 4 * oq;
    auto post = [&](int s, int t) {  // helper thread 0 only; ticket t = item * ksplit + ks
      const int item = SPLITK ? t / ksplit : t, ks = SPLITK ? t - item * ksplit : 0;
      const int tile = item >> pshift;
      const int row = tile / a.tiles_x;
      const intx4 v = {item, tile - row * a.tiles_x, (row % a.tiles_y) | (ks << 16), row / a.tiles_y};
      *(intx4*)(slots + 4 * s) = v;
    };
    // Input side.  The 18x10-pixel window of the tile (halo 1; 16 channels of the chunk, fp32) is
    // loaded ONCE from global memory - 720 float4 for 256 threads - and staged in LDS; every
    // thread then reads its 3x4 patch from there.  (Loading the overlapping patches straight from
    // global memory costs 12 loads per thread and chunk; the vector-memory path, not the MFMA, was
    // then what bounded the kernel: -30 % kernel time with those loads removed.)
    //   R[s]      window registers: the chunk that will be staged into window buffer s
    //   phase c:  stage W[c&1] <- R[c&1] (chunk c+2); commit target c+1 from W[(c+1)&1] to V[(c+1)&1];
    //             barrier; issue R[c&1] <- chunk c+4
    // so a global load has two phases to arrive, a staged window one barrier to become visible.
    struct WinAim {
      unsigned off[3];   // byte offsets of this thread's (pixel, quad) items inside the image (out of range = 0)
      unsigned coff;     // byte offset of the work item's first chunk inside a pixel
      __amdgpu_buffer_rsrc_t rsrc;
    };
    WinAim am;
    unsigned stage_off[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int i = ht + 256 * j;
      stage_off[j] = (unsigned)((i >> 2) * (CK * 4) + (i & 3) * 16);
    }
    auto aim = [&](WinAim& m, const Tile& t) {
      m.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (size_t)t.n * a.H * a.W * Cin), 0,
                                                 a.H * a.W * Cin * 4, 0x00020000);
      m.coff = (unsigned)(t.ks * nchunks * (CK * 4));
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int i = ht + 256 * j;
        const int pix = i >> 2, q = i & 3;
        const int wy = pix / WINW, wx = pix - wy * WINW;
        const int gy = t.ty * 8 - 1 + wy, gx = t.tx * 16 - 1 + wx;
        const bool ok = i < WINW * WINH * 4 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        m.off[j] = ok ? (unsigned)(((gy * a.W + gx) * Cin + 4 * q) * 4) : 0x80000000u;
      }
    };
    auto aim_nowhere = [&](WinAim& m) {
      m.coff = 0;
#pragma unroll
      for (int j = 0; j < 3; ++j) m.off[j] = 0x80000000u;
    };
    auto issue = [&](floatx4 (&r)[3], int chunk) {
      if (WB_ABL & 2) return;
#pragma unroll
      for (int j = 0; j < 3; ++j)
        r[j] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(am.rsrc, am.off[j], am.coff + chunk * (CK * 4), 0));
    };
    auto stage = [&](const floatx4 (&r)[3], int wbuf) {
      unsigned char* wb = lds_raw + STAGEOFF + wbuf * WINB;
#pragma unroll
      for (int j = 0; j < 3; ++j)
        if (j < 2 || ht < WINW * WINH * 4 - 512) *(floatx4*)(wb + stage_off[j]) = r[j];
    };
    // this thread's patch inside a staged window: rows 2*pty+hx .. +2, columns 2*ptx .. +3, quad pq
    const int patch0 = ((2 * pty + hx) * WINW + 2 * ptx) * (CK * 4) + pq * 16;
    // fp32 value -> bf16 planes: hi = bf16(v), mid = bf16(v - hi) (, lo = bf16(v - hi - mid))
    auto split_store = [&](floatx4 v, unsigned char* dst) {
      const unsigned h0 = cvt_pk_bf16(v[0], v[1]), h1 = cvt_pk_bf16(v[2], v[3]);
      *(uintx2*)dst = uintx2{h0, h1};
      float r0 = v[0] - bf16_lo(h0), r1 = v[1] - bf16_hi(h0), r2 = v[2] - bf16_lo(h1), r3 = v[3] - bf16_hi(h1);
      const unsigned m0 = cvt_pk_bf16(r0, r1), m1 = cvt_pk_bf16(r2, r3);
      *(uintx2*)(dst + PLANEB) = uintx2{m0, m1};
      if (P == 3) {
        r0 -= bf16_lo(m0); r1 -= bf16_hi(m0); r2 -= bf16_lo(m1); r3 -= bf16_hi(m1);
        *(uintx2*)(dst + 2 * PLANEB) = uintx2{cvt_pk_bf16(r0, r1), cvt_pk_bf16(r2, r3)};
      }
    };
    // B^T d B for this thread's two xi rows: window buffer wbuf -> V buffer vb.  zxi / znu: the
    // structurally zero Winograd row / column of this chunk (4 = none; see the MFMA waves): those
    // positions are never read, so they are neither transformed nor split nor stored.
    auto commit = [&](int wbuf, int vb, unsigned zxi, unsigned znu) {
      if (WB_ABL & 4) return;
      const unsigned char* src = lds_raw + STAGEOFF + wbuf * WINB + patch0;
      unsigned char* dst = vmine + vb * VBUFB;
      const bool lo_live = zxi != (unsigned)(2 * hx), hi_live = zxi != (unsigned)(2 * hx + 1);   // wave-uniform
      floatx4 xlo[4], xhi[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const floatx4 ra = *(const floatx4*)(src + (0 * WINW + k) * (CK * 4));
        const floatx4 rb = *(const floatx4*)(src + (1 * WINW + k) * (CK * 4));
        const floatx4 rc = *(const floatx4*)(src + (2 * WINW + k) * (CK * 4));
        if (hx == 0) {   // wave-uniform
          xlo[k] = ra - rc;   // xi 0: r0 - r2
          xhi[k] = rb + rc;   // xi 1: r1 + r2
        } else {
          xlo[k] = rb - ra;   // xi 2: r2 - r1
          xhi[k] = ra - rc;   // xi 3: r1 - r3
        }
      }
      // columns: nu 0: x0-x2, 1: x1+x2, 2: x2-x1, 3: x1-x3
      if (lo_live) {
        if (znu != 0) split_store(xlo[0] - xlo[2], dst + 0 * POSB);
        split_store(xlo[1] + xlo[2], dst + 1 * POSB);
        split_store(xlo[2] - xlo[1], dst + 2 * POSB);
        if (znu != 3) split_store(xlo[1] - xlo[3], dst + 3 * POSB);
      }
      if (hi_live) {
        if (znu != 0) split_store(xhi[0] - xhi[2], dst + 4 * POSB);
        split_store(xhi[1] + xhi[2], dst + 5 * POSB);
        split_store(xhi[2] - xhi[1], dst + 6 * POSB);
        if (znu != 3) split_store(xhi[1] - xhi[3], dst + 7 * POSB);
      }
    };
    // zero row / column of tile-local chunk k of work item `item` (MODE 1: by channel block; MODE 2: by phase)
    auto zero_of = [&](const Tile& t, int k, unsigned& zxi, unsigned& znu) {
      zxi = 4;
      znu = 4;
      if (MODE == 1) {
        const int blk = (t.ks * nchunks + k) / (tchunks >> 2);
        if (blk >> 1) zxi = 3;
        if (blk & 1) znu = 3;
      } else if (MODE == 2) {
        const int phase = t.item & 3;
        if (phase >> 1) zxi = 0;
        if (phase & 1) znu = 0;
      }
    };
    struct OutAim {
      unsigned po[4];
      __amdgpu_buffer_rsrc_t rs;
    };
    auto next_ticket = [&]() { return (int)(atomicAdd(a.ticket, 1ULL) + gridDim.x); };
    auto aim_out = [&](OutAim& o, const Tile& t) {
      const int phase = t.item & (a.nphase - 1);
      const int ppy = phase >> 1, ppx = phase & 1;
      const int OH = a.nphase == 4 ? 2 * a.H : a.H, OW = a.nphase == 4 ? 2 * a.W : a.W;
      o.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (SPLITK ? (size_t)t.ks * a.part_stride : (size_t)0) +
                                                       (size_t)t.n * OH * OW * a.ostride + a.ooff), 0,
                                               (OH * OW * a.ostride - a.ooff) * 4, 0x00020000);
#pragma unroll
      for (int ij = 0; ij < 4; ++ij) {
        const int oy = t.ty * 8 + 2 * oty + (ij >> 1);
        const int ox = t.tx * 16 + 2 * otx + (ij & 1);
        const unsigned po =
            (unsigned)(a.nphase == 4 ? ((2 * oy + ppy) * OW + (2 * ox + ppx)) * a.ostride
                       : a.s2d       ? ((oy >> 1) * (a.W >> 1) + (ox >> 1)) * (4 * a.Cout) + ((oy & 1) * 2 + (ox & 1)) * a.Cout
                                     : (oy * a.W + ox) * a.ostride) * 4u + 16u * oq;
        o.po[ij] = oy < a.H && ox < a.W ? po : 0x80000000u;
      }
    };
    // one 32-channel group of the finished outputs: 4 LDS reads + 4 buffer stores per thread.  The
    // groups of a tile are copied out one per chunk pair of the NEXT tile: the vector-memory path
    // issues a 1 KB store instruction in >= 16 cycles, and all 64 KB behind one barrier would hold
    // the MFMA waves for thousands of cycles.
    // No branch inside: a store whose group or channel quad lies beyond Cout gets the out-of-range
    // offset the hardware drops.  (The compiler's s_waitcnt placement is exact only when every path
    // through a phase issues the same memory operations: with a conditional copy-out the wait for
    // the window loads turned into a wait for the stores issued just before - 700 cycles per phase.)
    auto store_group = [&](const OutAim& o, int g) {
      const float* src = yreg + yread;
      const bool chan_ok = g * 32 + 4 * oq < a.Cout;
#pragma unroll
      for (int ij = 0; ij < 4; ++ij) {
        const floatx4 v = *(const floatx4*)(src + (ij * 4 + (g & 3)) * 32 * WP);
        const unsigned off = chan_ok ? o.po[ij] : 0x80000000u;
        if (NT_OUT)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, v), o.rs, off, g * 128, 2);
        else
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, v), o.rs, off, g * 128, 0);
      }
    };

    if (ht == 0) {
      post(0, (int)blockIdx.x);
      post(1, next_ticket());
    }
    __syncthreads();  // P0
    Tile cur = read_slot(0);
    floatx4 R0[3], R1[3];
    int ticket_pre = a.ntiles * ksplit;
    aim(am, cur);
    issue(R0, 0);
    issue(R1, 1);
    stage(R0, 0);
    stage(R1, 1);
    issue(R0, 2);
    issue(R1, 3);
    if (ht == 0 && read_slot(1).item < a.ntiles) ticket_pre = next_ticket();
    __syncthreads();  // P1: windows of chunks 0 and 1 are staged
    {
      unsigned zxi, znu;
      zero_of(cur, 0, zxi, znu);
      commit(0, 0, zxi, znu);    // V[0] = (cur, 0)
    }
    __syncthreads();  // P
    int s_nxt = 1, s_wr = 2;
    OutAim oa;
    oa.rs = am.rsrc;
#pragma unroll
    for (int ij = 0; ij < 4; ++ij) oa.po[ij] = 0x80000000u;
    int tile_count = 0;
    (void)tile_count;
    while (cur.item < a.ntiles) {
      const Tile nxt = read_slot(s_nxt);
      const bool more = nxt.item < a.ntiles;
      tile_count++;
      // One phase (tile-local chunk c; targets past the last chunk belong to the next tile).  Every
      // phase issues the same memory operations whatever the tile: with no next tile the aim points
      // nowhere (loads return 0 without traffic) and the staged / committed chunks are never read.
      auto phase = [&](floatx4 (&R)[3], int c) {
        WSTAMPC(8, 4 * (c - WB_STAMP_C0), 64);
        stage(R, c & 1);                                                 // window of chunk c+2
        WSTAMPC(8, 4 * (c - WB_STAMP_C0) + 1, 64);
        {
          unsigned zxi, znu;                                             // target c+1
          if (c + 1 < nchunks) zero_of(cur, c + 1, zxi, znu); else zero_of(nxt, 0, zxi, znu);
          commit((c + 1) & 1, (c + 1) & 1, zxi, znu);
        }
        WSTAMPC(8, 4 * (c - WB_STAMP_C0) + 2, 64);
        __syncthreads();  // B_c
        WSTAMPC(8, 4 * (c - WB_STAMP_C0) + 3, 64);
        const int k4 = c + 4;
        if (k4 == nchunks) {   // phase n-4: from here on every load is for the next tile
          if (more) aim(am, nxt); else aim_nowhere(am);
        }
        issue(R, k4 < nchunks ? k4 : k4 - nchunks);
      };
      if (ht == 0 && more) post(s_wr, ticket_pre);
      // outputs of the previous tile (Y is rewritten at this tile's fold): one 32-channel group per chunk
      // pair in the first eight phases (both groups of a pair at once when there are only four phases)
      const int npair_out = nchunks >= 8 ? 4 : 2;
      for (int c = 0; c < 2 * npair_out; c += 2) {
        store_group(oa, nchunks >= 8 ? (c >> 1) : c);
        if (nchunks < 8) store_group(oa, c + 1);
        phase(R0, c);
        phase(R1, c + 1);
      }
      for (int c = 2 * npair_out; c < nchunks; c += 2) {
        phase(R0, c);
        phase(R1, c + 1);
      }
      if (ht == 0 && more && read_slot(s_wr).item < a.ntiles) ticket_pre = next_ticket();
      aim_out(oa, cur);
      __syncthreads();  // E1
      __syncthreads();  // E2
      cur = nxt;
      const int s_old = s_nxt;
      s_nxt = s_wr;
      s_wr = s_old == 0 ? 2 : s_old - 1;
    }
    for (int g = 0; g < 4; ++g) store_group(oa, g);   // outputs of the last tile (dropped offsets if there was none)
#if WB_STAMP
    if (wave == 8) {
      wb_stamps[blockIdx.x * 128 + lane] = ((long long*)(lds_raw + STAMPOFF))[lane];
      wb_stamps[blockIdx.x * 128 + 64 + lane] = ((long long*)(lds_raw + STAMPOFF))[64 + lane];
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
  const int nt = wave & 3, ph = wave >> 2;
  const bool nvalid = nt * 32 < a.CoutP;
  constexpr bool ZSKIP = MODE != 0;
  // U stream: [pos][chunk][plane][CoutP][16 bf16]; a fragment = 64 lanes x 16 bytes
  const unsigned plane_b = (unsigned)a.CoutP * 32u;
  const unsigned chunk_b = plane_b * (unsigned)P;
  const unsigned pos_b = chunk_b * (unsigned)tchunks;
  const unsigned ulane = (unsigned)((((nvalid ? nt : 0) * 32 + l31) * 2 + h) * 16);
  // V reads: row l31 of position ph*8+p, k-block h (swizzled)
  const int aread = l31 * ROWB + ((h ^ ((l31 >> 3) & 1)) << 4);
  float pbias = 0.f, pbeta = 1.f, pgamma = 0.f;
  {
    const int col = nt * 32 + l31;
    if (col < a.Cout) {
      pbias = SPLITK ? 0.f : a.bias[col];
      if (a.act == DSIC_ACT_GDN || a.act == DSIC_ACT_IGDN) {
        pbeta = a.beta[col];
        pgamma = a.gamma[col];
      }
    }
  }
  // The loop is compiled once per position half: the positions of a step are then compile-time LDS offsets.
  auto mfma_waves = [&](auto ph_tag) {
  constexpr int PH = decltype(ph_tag)::value;
  constexpr unsigned ptab = PosTab<MODE, PH>::value;
  auto pos_of = [](int pi) { return (int)((ptab >> (4 * pi)) & 15u); };   // global position xi*4 + nu of step pi
  floatx16 acc[8];
  bf16x8 Bq[RING][P];
  __syncthreads();  // P0
  __syncthreads();  // P1
  __syncthreads();  // P
  Tile cur = read_slot(0);
  const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)a.u, 0, (int)(16u * pos_b * (unsigned)a.nphase), 0x00020000);
  // fetch stream position: (tile's phase, chunk, position index pi) -> scalar byte offset
  // (+ the work item's first chunk)
  auto soff_item = [&](const Tile& t) {
    return (unsigned)(t.item & (a.nphase - 1)) * (unsigned)a.u_phase_bytes + (unsigned)(t.ks * nchunks) * chunk_b;
  };
  unsigned soff_phase = soff_item(cur);
  auto soff_of = [&](unsigned phase_off, int chunk, int pi) {
    return phase_off + (unsigned)pos_of(pi) * pos_b + (unsigned)chunk * chunk_b;
  };
  auto fetch = [&](bf16x8 (&dst)[P], unsigned so) {
    if (WB_ABL & 1) so = 0;
#pragma unroll
    for (int q = 0; q < P; ++q)
      dst[q] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(urs, ulane, so + (unsigned)q * plane_b, 0));
  };
#pragma unroll
  for (int f = 0; f < RING; ++f) fetch(Bq[f], soff_of(soff_phase, 0, f));
  int s_nxt = 1;
  int tile_count = 0;
  (void)tile_count;
  while (cur.item < a.ntiles) {
    tile_count++;
    const Tile nxt = read_slot(s_nxt);
    const unsigned soff_phase_nxt = soff_item(nxt.item < a.ntiles ? nxt : cur);
    auto chunk_body = [&](auto first_tag, int chunk) {
      constexpr bool FIRST = decltype(first_tag)::value;
      const bool last = chunk + 1 == nchunks;
      WSTAMPC(0, 3 * (chunk - WB_STAMP_C0), 60);
      unsigned zero_xi = 4, zero_nu = 4;
      if (MODE == 1) {
        const int blk = (cur.ks * nchunks + chunk) / (tchunks >> 2);
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
#if WB_ADOUBLE
      // two sets of V fragments: the reads of step pi+1 are issued before the MFMAs of step pi
      bf16x8 Aq[2][P];
#pragma unroll
      for (int q = 0; q < P; ++q) Aq[0][q] = *(const bf16x8*)(vb + pos_of(0) * POSB + q * PLANEB);
#define AQ(q) Aq[pi & 1][q]
#else
      // one set of V fragments: the reads of step pi+1 are issued right behind the MFMAs of step pi
      // (they land long after those MFMAs have read their sources); the second MFMA wave of the
      // SIMD covers the LDS latency
      bf16x8 Aq1[P];
#pragma unroll
      for (int q = 0; q < P; ++q) Aq1[q] = *(const bf16x8*)(vb + pos_of(0) * POSB + q * PLANEB);
#define AQ(q) Aq1[q]
#endif
#pragma unroll
      for (int pi = 0; pi < 8; ++pi) {
        const int p0 = pi;   // accumulators are indexed by step
        const bool live = !ZSKIP || !is_zero(pi);  // wave-uniform
#if WB_ADOUBLE
        if (pi + 1 < 8) {
#pragma unroll
          for (int q = 0; q < P; ++q) Aq[(pi + 1) & 1][q] = *(const bf16x8*)(vb + pos_of(pi + 1) * POSB + q * PLANEB);
        }
#endif
        if (live && !(WB_ABL & 16)) {
          const floatx16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          floatx16 c = FIRST ? zero : acc[p0];
          if (P == 3) {  // small terms first
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AQ(0), Bq[pi % RING][P - 1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AQ(P - 1), Bq[pi % RING][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AQ(1), Bq[pi % RING][1], c, 0, 0, 0);
          }
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AQ(0), Bq[pi % RING][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AQ(1), Bq[pi % RING][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AQ(0), Bq[pi % RING][0], c, 0, 0, 0);
          if (WB_ABL & 128) {   // ablation 128: a second M tile per U fragment (V re-read, 3 more MFMAs)
#pragma unroll
            for (int q = 0; q < P; ++q) Aq1[q] = *(const bf16x8*)(vb + pos_of(pi) * POSB + q * PLANEB + 8 * ROWB);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AQ(0), Bq[pi % RING][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AQ(1), Bq[pi % RING][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AQ(0), Bq[pi % RING][0], c, 0, 0, 0);
          }
          acc[p0] = c;
        } else if (FIRST) {
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[p0][e] = 0.f;
        }
#if !WB_ADOUBLE
        if (pi + 1 < 8 && !((WB_ABL & 32) && (pi & 1) == 0)) {   // ablation 32: every second V read dropped
#pragma unroll
          for (int q = 0; q < P; ++q) Aq1[q] = *(const bf16x8*)(vb + pos_of(pi + 1) * POSB + q * PLANEB);
        }
#endif
        {
          // refill this ring slot with the fragments of step pi + RING (next chunk / next tile past 7)
          const int f = pi + RING;
          const unsigned so = f < 8 ? soff_of(soff_phase, chunk, f)
                                    : (last ? soff_of(soff_phase_nxt, 0, f - 8) : soff_of(soff_phase, chunk + 1, f - 8));
          // a structurally zero position of THIS chunk is never multiplied: its fragments are not
          // fetched at all (the vector-memory path, 64 B/clk per CU, is what bounds this kernel).
          // Steps of the next chunk / tile (f >= 8) are fetched unconditionally.
          if (!(ZSKIP && f < 8 && is_zero(f)) && !((WB_ABL & 64) && (pi & 1))) fetch(Bq[pi % RING], so);   // ablation 64: half of the U fetches
        }
      }
#undef AQ
      WSTAMPC(0, 3 * (chunk - WB_STAMP_C0) + 1, 60);
      __syncthreads();  // B_chunk
      WSTAMPC(0, 3 * (chunk - WB_STAMP_C0) + 2, 60);
    };
    chunk_body(std::true_type{}, 0);
    for (int chunk = 1; chunk < nchunks; ++chunk) chunk_body(std::false_type{}, chunk);
    soff_phase = soff_phase_nxt;

    // ---- inverse transform (as conv_wino.hip) -----------------------------------------------
    WSTAMP(0, 60);
    if (WB_ABL & 8) {
      __syncthreads();
    } else {
      float* yown = yreg + ((4 * (2 * PH) + nt) * 32 + 4 * h) * WP + l31;
      float* yoth = yreg + ((4 * (2 * (PH ^ 1)) + nt) * 32 + 4 * h) * WP + l31;
      // Y = A^T M A: this wave's 8 positions contribute to all four outputs (i, j) of a tile.  It owns output
      // row i = ph (written to its half of the output region) and sends its share of row ph^1 to the other
      // position half, which finishes that row after the barrier.
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
      WSTAMP(0, 61);
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
    }
    WSTAMP(0, 62);
    __syncthreads();  // E2
    WSTAMP(0, 63);
    cur = nxt;
    s_nxt = s_nxt == 2 ? 0 : s_nxt + 1;
  }
  };
  if (ph == 0)
    mfma_waves(std::integral_constant<int, 0>{});
  else
    mfma_waves(std::integral_constant<int, 1>{});
}

// fp32 transformed weights [16][Cin/8][CoutP][8] (dsic_pack_wino_*_weight, one block per phase) ->
// bf16 planes [16][Cin/16][PLANES][CoutP][16]
__global__ void split_u_kernel(const float* __restrict__ u32, unsigned short* __restrict__ dst, int Cin, int CoutP,
                               int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one per (pos, chunk, n, k)
  if (i >= total) return;
  const int k = i & 15;
  int64_t r = i >> 4;
  const int n = r % CoutP;
  r /= CoutP;
  const int nchunks = Cin / CK;
  const int chunk = r % nchunks;
  const int pos = r / nchunks;
  const int c = chunk * CK + k;
  const float v = u32[(((size_t)pos * (Cin / 8) + (c >> 3)) * CoutP + n) * 8 + (c & 7)];
  float rem = v;
#pragma unroll
  for (int q = 0; q < P; ++q) {
    const unsigned pk = cvt_pk_bf16(rem, 0.f);
    dst[((((size_t)pos * nchunks + chunk) * P + q) * CoutP + n) * 16 + k] = (unsigned short)(pk & 0xFFFFu);
    rem = rem - bf16_lo(pk);
  }
}

// split-K epilogue: out = act(sum_ks part[ks] + bias), partial sums added in the fixed order ks = 0..S-1.
// One thread per (pixel, channel quad); the partial buffers use the output's own addressing.
template <int ACT>
__global__ void splitk_reduce_kernel(const float* __restrict__ part, int S, int64_t part_stride, float* __restrict__ out,
                                     int64_t npix, int C, int cmod, int ostride, int ooff,
                                     const float* __restrict__ bias, const float* __restrict__ beta,
                                     const float* __restrict__ gamma) {
  const int q4 = C >> 2;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * q4) return;
  const int64_t pix = i / q4;
  const int c = (int)(i - pix * q4) * 4;
  const int64_t off = pix * ostride + ooff + c;
  floatx4 v = *(const floatx4*)(part + off);
  for (int ks = 1; ks < S; ++ks) v = v + *(const floatx4*)(part + (int64_t)ks * part_stride + off);
  const int cb = c % cmod;
  v = v + *(const floatx4*)(bias + cb);
  if (ACT == DSIC_ACT_GDN || ACT == DSIC_ACT_IGDN) {
    const floatx4 be = *(const floatx4*)(beta + cb), ga = *(const floatx4*)(gamma + cb);
    const floatx2 lo = gdn_pair<ACT == DSIC_ACT_IGDN>(floatx2{v[0], v[1]}, floatx2{be[0], be[1]}, floatx2{ga[0], ga[1]});
    const floatx2 hi = gdn_pair<ACT == DSIC_ACT_IGDN>(floatx2{v[2], v[3]}, floatx2{be[2], be[3]}, floatx2{ga[2], ga[3]});
    v = floatx4{lo[0], lo[1], hi[0], hi[1]};
  } else if (ACT == DSIC_ACT_RELU) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
  }
  *(floatx4*)(out + off) = v;
}

}  // namespace wb
}  // namespace dsic

using namespace dsic;

#if WB_STAMP
extern "C" int dsic_debug_wb_stamps(long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(wb::wb_stamps), sizeof(long long) * 256 * 128) == hipSuccess ? 0 : 2;
}
#endif

extern "C" int dsic_wino_bf16_planes(void) { return wb::P; }

extern "C" int64_t dsic_wino_bf16_weight_bytes(int Cout, int Cin) {
  return (int64_t)16 * (Cin / wb::CK) * wb::P * round_up(Cout, 32) * 16 * 2;
}

extern "C" int dsic_split_wino_weight_bf16(const float* u_f32, void* dst, int Cout, int Cin, int nphase,
                                           void* stream) {
  DSIC_REQUIRE(u_f32 && dst && Cout > 0 && Cin > 0 && Cin % 32 == 0 && (nphase == 1 || nphase == 4),
               "split_wino_weight_bf16: bad argument");
  const int CoutP = round_up(Cout, 32);
  const int64_t total = (int64_t)16 * Cin * CoutP;
  const int64_t fstride = (int64_t)16 * (Cin / 8) * CoutP * 8;
  for (int ph = 0; ph < nphase; ++ph)
    hipLaunchKernelGGL(wb::split_u_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       u_f32 + (size_t)ph * fstride,
                       (unsigned short*)((unsigned char*)dst + (size_t)ph * dsic_wino_bf16_weight_bytes(Cout, Cin)), Cin,
                       CoutP, total);
  return check_launch("split_wino_weight_bf16");
}

int dsic_wbm_launch(wb::Args& a, hipStream_t st);   // conv_wino_bf16m.hip

static int wb_launch(wb::Args& a, hipStream_t st) {
  const int B = a.B, H = a.H, W = a.W;
  a.tiles_x = ceil_div(W, 16);
  a.tiles_y = ceil_div(H, 8);
  const int64_t nt = (int64_t)a.tiles_x * a.tiles_y * B * a.nphase;
  if (a.ksplit < 1) a.ksplit = 1;
  a.kchunks = a.Cin / wb::CK / a.ksplit;
  DSIC_REQUIRE(a.kchunks * a.ksplit * wb::CK == a.Cin && a.kchunks >= 4 && a.kchunks % 2 == 0 && a.ksplit < 256,
               "conv_wino_bf16: ksplit=%d does not divide Cin=%d into even runs of >= 4 chunks", a.ksplit, a.Cin);
  DSIC_REQUIRE(nt * a.ksplit < ((int64_t)1 << 31), "conv_wino_bf16: too many tiles");
  if (a.ostride <= 0) a.ostride = a.Cout;
  DSIC_REQUIRE(a.ooff >= 0 && a.ooff % 4 == 0 && a.ostride % 4 == 0 && a.ooff + a.Cout <= a.ostride,
               "conv_wino_bf16: output slice [%d, %d) does not fit a pixel stride of %d channels", a.ooff,
               a.ooff + a.Cout, a.ostride);
  DSIC_REQUIRE(a.ostride == a.Cout || !a.s2d, "conv_wino_bf16: a Cout slice cannot be stored space-to-depth");
  DSIC_REQUIRE((int64_t)H * W * a.Cin * 4 < ((int64_t)1 << 31) &&
                   (int64_t)H * W * a.ostride * 4 * (a.nphase == 4 ? 4 : 1) < ((int64_t)1 << 31),
               "conv_wino_bf16: one image must stay below 2 GiB (32-bit offsets inside an image)");
  DSIC_REQUIRE(a.u_phase_bytes * a.nphase < ((int64_t)1 << 31), "conv_wino_bf16: transformed weights must stay below 2 GiB");
  if (a.ksplit == 1 && dsic_wino_bf16_m64(H, W, a.Cin, a.nphase)) return dsic_wbm_launch(a, st);   // large layers: conv_wino_bf16m.hip
  a.ntiles = (int)nt;
  a.nt_out = a.ksplit == 1 && (int64_t)B * H * W * a.Cout * 4 * (a.nphase == 4 ? 4 : 1) > (300ll << 20);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  static bool attr_set[64] = {};
  if (dev < 0 || dev >= 64) dev = 0;
  if (!attr_set[dev]) {
    const void* fns[8] = {(const void*)wb::conv_wino_bf16_kernel<0, false, false>, (const void*)wb::conv_wino_bf16_kernel<1, false, false>,
                          (const void*)wb::conv_wino_bf16_kernel<2, false, false>, (const void*)wb::conv_wino_bf16_kernel<0, true, false>,
                          (const void*)wb::conv_wino_bf16_kernel<1, true, false>,  (const void*)wb::conv_wino_bf16_kernel<2, true, false>,
                          (const void*)wb::conv_wino_bf16_kernel<0, false, true>,  (const void*)wb::conv_wino_bf16_kernel<1, false, true>};
    for (int i = 0; i < 8; ++i) {
      const hipError_t e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, wb::LDS_TOTAL);
      if (e != hipSuccess) {
        set_error("conv_wino_bf16: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return DSIC_EHIP;
      }
    }
    attr_set[dev] = true;
  }
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
  const int64_t nwork = (int64_t)a.ntiles * a.ksplit;
  const int grid = nwork < max_grid ? (int)nwork : max_grid;
#define WB_LAUNCH(M)                                                                                          \
  do {                                                                                                        \
    if (a.nt_out)                                                                                             \
      hipLaunchKernelGGL((wb::conv_wino_bf16_kernel<M, true, false>), dim3(grid), dim3(wb::THREADS), wb::LDS_TOTAL, st, a);  \
    else                                                                                                      \
      hipLaunchKernelGGL((wb::conv_wino_bf16_kernel<M, false, false>), dim3(grid), dim3(wb::THREADS), wb::LDS_TOTAL, st, a); \
  } while (0)
  if (a.ksplit > 1) {
    DSIC_REQUIRE(a.nphase == 1, "conv_wino_bf16: split-K exists for the 3x3 layers only");
    if (a.s2d_in)
      hipLaunchKernelGGL((wb::conv_wino_bf16_kernel<1, false, true>), dim3(grid), dim3(wb::THREADS), wb::LDS_TOTAL, st, a);
    else
      hipLaunchKernelGGL((wb::conv_wino_bf16_kernel<0, false, true>), dim3(grid), dim3(wb::THREADS), wb::LDS_TOTAL, st, a);
  } else if (a.s2d_in)
    WB_LAUNCH(1);
  else if (a.nphase == 4)
    WB_LAUNCH(2);
  else
    WB_LAUNCH(0);
#undef WB_LAUNCH
  return check_launch("conv_wino_bf16");
}

extern "C" int dsic_conv3x3_wino_bf16_nhwc(const float* in, const void* u_planes, const float* bias,
                                           const float* beta, const float* gamma, float* out, int B, int H,
                                           int W, int Cin, int Cout, int act, int s2d_out, int s2d_in,
                                           int out_cstride, int out_coff, void* ticket, void* stream) {
  DSIC_REQUIRE(in && u_planes && bias && out && ticket, "conv3x3_wino_bf16: null pointer");
  DSIC_REQUIRE(B > 0 && H > 0 && W > 0, "conv3x3_wino_bf16: empty tensor");
  DSIC_REQUIRE(Cin >= 64 && Cin % 32 == 0, "conv3x3_wino_bf16: Cin=%d must be a multiple of 32, >= 64", Cin);
  DSIC_REQUIRE(Cout > 0 && Cout % 4 == 0 && Cout <= 128, "conv3x3_wino_bf16: Cout=%d must be a multiple of 4, <= 128", Cout);
  DSIC_REQUIRE(act >= 0 && act <= 3, "conv3x3_wino_bf16: act=%d", act);
  DSIC_REQUIRE(!(act == DSIC_ACT_GDN || act == DSIC_ACT_IGDN) || (beta && gamma), "conv3x3_wino_bf16: GDN needs beta and gamma");
  DSIC_REQUIRE(!s2d_out || (H % 2 == 0 && W % 2 == 0), "conv3x3_wino_bf16: space-to-depth output needs even H and W");
  DSIC_REQUIRE(!s2d_in || Cin % 128 == 0, "conv3x3_wino_bf16: space-to-depth input needs Cin = 4*Cs with Cs %% 32 == 0");
  wb::Args a{};
  a.in = in; a.u = u_planes; a.bias = bias; a.beta = beta; a.gamma = gamma; a.out = out;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.CoutP = round_up(Cout, 32); a.act = act;
  a.s2d = s2d_out; a.s2d_in = s2d_in ? 1 : 0;
  a.ostride = out_cstride; a.ooff = out_coff;
  a.ticket = (unsigned long long*)ticket;
  a.nphase = 1; a.u_phase_bytes = dsic_wino_bf16_weight_bytes(Cout, Cin);
  return wb_launch(a, (hipStream_t)stream);
}

// Split-K policy, a function of the layer's geometry only (never of the batch size: a patch's bits must not depend
// on the batch it is in): images of fewer than four 16x8-pixel tiles split their input channels over up to four
// work items per tile (4 / tiles per image), each with an even run of >= 4 chunks.  (8 / tiles per image was
// measured too: at 64 images per batch the second doubling only adds folds and partial-sum traffic.)
extern "C" int dsic_wino_bf16_ksplit(int H, int W, int Cin) {
  if (H <= 0 || W <= 0 || Cin < 64 || Cin % 32) return 1;
  const int t_img = ceil_div(W, 16) * ceil_div(H, 8);
  int S = t_img >= 4 ? 1 : 4 / t_img;
  const int nchunks = Cin / wb::CK;
  while (S > 1 && (nchunks % S || nchunks / S < 4 || (nchunks / S) % 2)) S >>= 1;
  return S;
}

extern "C" int dsic_conv3x3_wino_bf16_splitk_nhwc(const float* in, const void* u_planes, const float* bias,
                                                  const float* beta, const float* gamma, float* out, int B, int H,
                                                  int W, int Cin, int Cout, int act, int s2d_out, int s2d_in,
                                                  int out_cstride, int out_coff, int ksplit, float* partials,
                                                  void* ticket, void* stream) {
  DSIC_REQUIRE(in && u_planes && bias && out && ticket && partials, "conv3x3_wino_bf16_splitk: null pointer");
  DSIC_REQUIRE(B > 0 && H > 0 && W > 0, "conv3x3_wino_bf16_splitk: empty tensor");
  DSIC_REQUIRE(Cin >= 64 && Cin % 32 == 0, "conv3x3_wino_bf16_splitk: Cin=%d must be a multiple of 32, >= 64", Cin);
  DSIC_REQUIRE(Cout > 0 && Cout % 4 == 0 && Cout <= 128, "conv3x3_wino_bf16_splitk: Cout=%d must be a multiple of 4, <= 128", Cout);
  DSIC_REQUIRE(act >= 0 && act <= 3, "conv3x3_wino_bf16_splitk: act=%d", act);
  DSIC_REQUIRE(!(act == DSIC_ACT_GDN || act == DSIC_ACT_IGDN) || (beta && gamma), "conv3x3_wino_bf16_splitk: GDN needs beta and gamma");
  DSIC_REQUIRE(!s2d_out || (H % 2 == 0 && W % 2 == 0), "conv3x3_wino_bf16_splitk: space-to-depth output needs even H and W");
  DSIC_REQUIRE(!s2d_in || Cin % 128 == 0, "conv3x3_wino_bf16_splitk: space-to-depth input needs Cin = 4*Cs with Cs %% 32 == 0");
  DSIC_REQUIRE(ksplit >= 2, "conv3x3_wino_bf16_splitk: ksplit=%d", ksplit);
  wb::Args a{};
  a.in = in; a.u = u_planes; a.bias = nullptr; a.beta = nullptr; a.gamma = nullptr; a.out = partials;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.CoutP = round_up(Cout, 32); a.act = DSIC_ACT_NONE;
  a.s2d = s2d_out; a.s2d_in = s2d_in ? 1 : 0;
  a.ostride = out_cstride; a.ooff = out_coff;
  a.ticket = (unsigned long long*)ticket;
  a.nphase = 1; a.u_phase_bytes = dsic_wino_bf16_weight_bytes(Cout, Cin);
  a.ksplit = ksplit;
  const int ostride = out_cstride > 0 ? out_cstride : Cout;
  a.part_stride = (int64_t)B * H * W * ostride;
  const int rc = wb_launch(a, (hipStream_t)stream);
  if (rc != DSIC_OK) return rc;
  // the space-to-depth output is a [B, H/2, W/2, 4*Cout] tensor: channel of the bias = index mod Cout
  const int64_t npix = s2d_out ? (int64_t)B * (H / 2) * (W / 2) : (int64_t)B * H * W;
  const int C = s2d_out ? 4 * Cout : Cout, rstride = s2d_out ? 4 * Cout : ostride;
  const int64_t nthreads = npix * (C / 4);
  const dim3 grid((unsigned)((nthreads + 255) / 256)), block(256);
#define WB_REDUCE(A)                                                                                               \
  hipLaunchKernelGGL((wb::splitk_reduce_kernel<A>), grid, block, 0, (hipStream_t)stream, (const float*)partials, ksplit, \
                     a.part_stride, out, npix, C, Cout, rstride, out_coff, bias, beta, gamma)
  if (act == DSIC_ACT_GDN) WB_REDUCE(DSIC_ACT_GDN);
  else if (act == DSIC_ACT_IGDN) WB_REDUCE(DSIC_ACT_IGDN);
  else if (act == DSIC_ACT_RELU) WB_REDUCE(DSIC_ACT_RELU);
  else WB_REDUCE(DSIC_ACT_NONE);
#undef WB_REDUCE
  return check_launch("conv3x3_wino_bf16_splitk");
}

extern "C" int dsic_conv_transpose2d_wino_bf16_nhwc(const float* in, const void* u_planes4, const float* bias,
                                                    const float* beta, const float* gamma, float* out, int B,
                                                    int H, int W, int Cin, int Cout, int act, void* ticket,
                                                    void* stream) {
  DSIC_REQUIRE(in && u_planes4 && bias && out && ticket, "convT_wino_bf16: null pointer");
  DSIC_REQUIRE(B > 0 && H > 0 && W > 0, "convT_wino_bf16: empty tensor");
  DSIC_REQUIRE(Cin >= 64 && Cin % 32 == 0, "convT_wino_bf16: Cin=%d must be a multiple of 32, >= 64", Cin);
  DSIC_REQUIRE(Cout > 0 && Cout % 4 == 0 && Cout <= 128, "convT_wino_bf16: Cout=%d must be a multiple of 4, <= 128", Cout);
  DSIC_REQUIRE(act >= 0 && act <= 3, "convT_wino_bf16: act=%d", act);
  DSIC_REQUIRE(!(act == DSIC_ACT_GDN || act == DSIC_ACT_IGDN) || (beta && gamma), "convT_wino_bf16: IGDN needs beta and gamma");
  wb::Args a{};
  a.in = in; a.u = u_planes4; a.bias = bias; a.beta = beta; a.gamma = gamma; a.out = out;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.CoutP = round_up(Cout, 32); a.act = act;
  a.ticket = (unsigned long long*)ticket;
  a.s2d = 0; a.s2d_in = 0; a.nphase = 4; a.u_phase_bytes = dsic_wino_bf16_weight_bytes(Cout, Cin);
  return wb_launch(a, (hipStream_t)stream);
}
