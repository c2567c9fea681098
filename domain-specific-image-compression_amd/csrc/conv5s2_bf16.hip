// conv(C, Cout, 5, stride 2) (code/modelv2/layers.py:54,60,65: g_a.2 / g_a.6 / g_a.10) as a DIRECT
// implicit GEMM on split-bf16 MFMAs, over the space-to-depth image its producer writes.
//
// Why not Winograd here (conv_wino_bf16.hip, MODE 1): with the bf16 matrix pipe 5.3x faster than the
// fp32-input one, F(2x2,3x3) over the 4*C-channel space-to-depth image is bound by the stream of
// transformed weights through the vector-memory path (16 positions x 4C x Cout x 4 bytes, 49/64 of
// them live = 3.2 MB per 128 output pixels for C = Cout = 128), not by the MFMAs (busy 0.23).  The
// direct form needs the 25 taps only (1.6 MB per 128 pixels) and no input transform at all; it pays
// with 2.04x the MFMA work, which that pipe has to spare.
//
// Geometry.  Output (oy, ox) = sum_{ky,kx,c} w[ky][kx][c] x[2oy+ky-2][2ox+kx-2][c].  With ky = 2u+a,
// kx = 2v+b (a, b in {0,1}) the input pixel is the space-to-depth pixel (oy+u-1, ox+v-1), channel
// block (a, b): per block a 3x3 (a=b=0), 3x2, 2x3 or 2x2 stride-1 correlation over C channels -
// 25 live (tap, block) pairs, no structural zero is ever multiplied.
//
// Workgroup = 768 threads on a 16x8-pixel output tile (M = 128 = four 32-row MFMA tiles, N = all
// Cout <= 128 channels = four 32-column tiles), persistent, tiles handed out by ticket:
//   * K walks 16-channel chunks (one MFMA k-step) of the 4C space-to-depth channels; per chunk the
//     18x10-pixel window is loaded once (3 float4 per helper thread, two chunks ahead), split into
//     the two bf16 planes (hi = bf16(v), mid = bf16(v - hi)) and staged in LDS as 80-byte pixel
//     records (16 hi | 16 mid | pad), double buffered.
//   * waves 0..7 (MFMA): wave (nt, kh) owns column tile nt, all four row tiles (64 accumulator VGPRs)
//     and every second tap of a chunk (kh = tap parity; the two partial sums meet in LDS at the end).
//     Per tap: 8 ds_read_b128 (4 row tiles x 2 planes at the tap's pixel offset), the tap's two
//     weight planes from a register ring loaded one chunk ahead, 12 MFMAs (hi*mid, mid*hi, hi*hi).
//   * waves 8..11 (helpers): window loads / split / staging, copy-out of the previous tile's outputs.
// Epilogue: kh=1 stores its partial sums into the output region [pixel][channel] of LDS, kh=0 adds
// its own, applies bias + GDN/ReLU in place; the helpers copy the tile to HBM 16 bytes per lane
// during the next tile's first chunks.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace dsic {
namespace c5 {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef int intx4 __attribute__((ext_vector_type(4)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
typedef unsigned uintx2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct Args {
  const float* in;   // space-to-depth image [B][H][W][4*Cs] (H, W = output size)
  const void* w;     // bf16 planes [blk 4][Cs/16][tap 9][plane 2][CoutP][16]
  const float* bias;
  const float* beta;
  const float* gamma;
  float* out;        // [B][H][W][Cout]
  int B, H, W, Cs, Cout, CoutP;
  int act;
  unsigned long long* ticket;
  int tiles_x, tiles_y, ntiles;
  int nt_out;
};

constexpr int CK = 16;
constexpr int WINW = 18, WINH = 10;
constexpr int REC = 80;                           // bytes per window pixel: 16 bf16 hi | 16 bf16 mid | pad
constexpr int ROWB = WINW * REC + 96;             // 1536: with this row pitch the 16-lane groups of a ds_read_b128 of an
                                                  // A fragment (two window rows x 16 pixels) hit 16 different 16-byte slots
constexpr int WINB = WINH * ROWB;                 // 15360
constexpr int YP = 132;                           // floats per output pixel row of the output region
constexpr int YOFF = 2 * WINB;                    // 30720
constexpr int YBYTES = 128 * YP * 4;              // 67584
constexpr int SLOTOFF = YOFF + YBYTES;
constexpr int LDS_TOTAL = SLOTOFF + 64;
constexpr int THREADS = 768;
constexpr int MAXT = 5;                           // taps of a chunk one wave can own (9 taps, every second)

struct Tile {
  int item, tx, ty, n;
};

__device__ __forceinline__ void split4(floatx4 v, uintx2& hi, uintx2& mid) {
  hi = __builtin_bit_cast(uintx2, __builtin_convertvector(v, bf16x4));
  floatx4 r;
  r[0] = v[0] - __builtin_bit_cast(float, hi[0] << 16);
  r[1] = v[1] - __builtin_bit_cast(float, hi[0] & 0xFFFF0000u);
  r[2] = v[2] - __builtin_bit_cast(float, hi[1] << 16);
  r[3] = v[3] - __builtin_bit_cast(float, hi[1] & 0xFFFF0000u);
  mid = __builtin_bit_cast(uintx2, __builtin_convertvector(r, bf16x4));
}

// taps of channel block blk = a*2+b: u in [0, 3-a), v in [0, 3-b); tap index t = u*nv + v
__device__ __forceinline__ int ntaps_of(int blk) { return (3 - (blk >> 1)) * (3 - (blk & 1)); }

template <bool NT_OUT>
__global__ __launch_bounds__(THREADS) void conv5s2_bf16_kernel(const Args a) {
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
    return t;
  };
  const int Cin = 4 * a.Cs;
  const int cpb = a.Cs / CK;          // chunks per channel block
  const int nchunks = 4 * cpb;        // >= 16 (host: Cs >= 64)

  if (wave >= 8) {
    // =================================== helper waves ===========================================
    const int ht = tid - 512;
    auto post = [&](int s, int item) {  // helper thread 0 only
      const int row = item / a.tiles_x;
      const intx4 v = {item, item - row * a.tiles_x, row % a.tiles_y, row / a.tiles_y};
      *(intx4*)(slots + 4 * s) = v;
    };
    struct WinAim {
      unsigned off[3];
      __amdgpu_buffer_rsrc_t rsrc;
    };
    WinAim am;
    unsigned stage_off[3];   // LDS record + quad of this thread's items
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int i = ht + 256 * j;
      stage_off[j] = (unsigned)(((i >> 2) / WINW) * ROWB + ((i >> 2) % WINW) * REC + (i & 3) * 8);
    }
    auto aim = [&](WinAim& m, const Tile& t) {
      m.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (size_t)t.n * a.H * a.W * Cin), 0,
                                                 a.H * a.W * Cin * 4, 0x00020000);
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
#pragma unroll
      for (int j = 0; j < 3; ++j) m.off[j] = 0x80000000u;
    };
    auto issue = [&](floatx4 (&r)[3], int chunk) {  // chunk = index of the 16-channel slice of the 4*Cs channels
#pragma unroll
      for (int j = 0; j < 3; ++j)
        r[j] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(am.rsrc, am.off[j], chunk * (CK * 4), 0));
    };
    auto stage = [&](const floatx4 (&r)[3], int wbuf) {
      unsigned char* wb = lds_raw + wbuf * WINB;
#pragma unroll
      for (int j = 0; j < 3; ++j)
        if (j < 2 || ht < WINW * WINH * 4 - 512) {
          uintx2 hi, mid;
          split4(r[j], hi, mid);
          *(uintx2*)(wb + stage_off[j]) = hi;
          *(uintx2*)(wb + stage_off[j] + 32) = mid;
        }
    };
    // copy-out: thread = (pixel op = idx>>5, channel quad = idx&31), idx = ht + 256*i, i < 16
    struct OutAim {
      unsigned po[16];
      __amdgpu_buffer_rsrc_t rs;
    };
    auto next_ticket = [&]() { return (int)(atomicAdd(a.ticket, 1ULL) + gridDim.x); };
    auto aim_out = [&](OutAim& o, const Tile& t) {
      o.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)t.n * a.H * a.W * a.Cout), 0,
                                               a.H * a.W * a.Cout * 4, 0x00020000);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int idx = ht + 256 * i;
        const int p = idx >> 5, q = idx & 31;
        const int oy = t.ty * 8 + (p >> 4), ox = t.tx * 16 + (p & 15);
        o.po[i] = (oy < a.H && ox < a.W && 4 * q < a.Cout) ? (unsigned)(((oy * a.W + ox) * a.Cout + 4 * q) * 4) : 0x80000000u;
      }
    };
    auto store_pair = [&](const OutAim& o, int i0) {   // two of the 16 float4 of this thread
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int i = i0 + k;
        const int idx = ht + 256 * i;
        const floatx4 v = *(const floatx4*)(yreg + (idx >> 5) * YP + (idx & 31) * 4);
        if (NT_OUT)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, v), o.rs, o.po[i], 0, 2);
        else
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, v), o.rs, o.po[i], 0, 0);
      }
    };

    if (ht == 0) {
      post(0, (int)blockIdx.x);
      post(1, next_ticket());
    }
    __syncthreads();  // P0
    Tile cur = read_slot(0);
    floatx4 R0[3], R1[3];
    int ticket_pre = a.ntiles;
    aim(am, cur);
    issue(R0, 0);
    issue(R1, 1);
    stage(R0, 0);              // W[0] = chunk 0
    issue(R0, 2);
    if (ht == 0 && read_slot(1).item < a.ntiles) ticket_pre = next_ticket();
    __syncthreads();  // P
    int s_nxt = 1, s_wr = 2;
    OutAim oa;
    oa.rs = am.rsrc;
#pragma unroll
    for (int i = 0; i < 16; ++i) oa.po[i] = 0x80000000u;
    while (cur.item < a.ntiles) {
      const Tile nxt = read_slot(s_nxt);
      const bool more = nxt.item < a.ntiles;
      // phase c: the MFMA waves consume W[c&1]; stage W[(c+1)&1] <- R (chunk c+1); barrier; issue R <- chunk c+3
      auto phase = [&](floatx4 (&R)[3], int c) {
        stage(R, (c + 1) & 1);
        __syncthreads();  // B_c
        const int k3 = c + 3;
        if (k3 == nchunks) {   // phase n-3: from here on every load is for the next tile
          if (more) aim(am, nxt); else aim_nowhere(am);
        }
        issue(R, k3 < nchunks ? k3 : k3 - nchunks);
      };
      if (ht == 0 && more) post(s_wr, ticket_pre);
      for (int c = 0; c < 16; c += 2) {      // the previous tile's outputs leave during the first 16 chunks
        store_pair(oa, c);
        phase(R1, c);
        phase(R0, c + 1);
      }
      for (int c = 16; c < nchunks; c += 2) {
        phase(R1, c);
        phase(R0, c + 1);
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
    for (int c = 0; c < 16; c += 2) store_pair(oa, c);   // outputs of the last tile
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
  const int nt = wave & 3, kh = wave >> 2;
  const bool nvalid = nt * 32 < a.CoutP;
  // weights: [blk][chunk-in-block][tap 9][plane 2][CoutP][16 bf16]; a fragment = 64 lanes x 16 bytes
  const unsigned plane_b = (unsigned)a.CoutP * 32u;
  const unsigned tap_b = 2u * plane_b;
  const unsigned chunk_b = 9u * tap_b;
  const unsigned ulane = (unsigned)((((nvalid ? nt : 0) * 32 + l31) * 2 + h) * 16);
  // window reads: output pixel of row tile mt, lane row r = l31: (2*mt + (r>>4), r&15); k-block h
  const int apix = (l31 >> 4) * ROWB + (l31 & 15) * REC + h * 16;
  float pbias = 0.f, pbeta = 1.f, pgamma = 0.f;
  {
    const int col = nt * 32 + l31;
    if (col < a.Cout) {
      pbias = a.bias[col];
      if (a.act == DSIC_ACT_GDN) {
        pbeta = a.beta[col];
        pgamma = a.gamma[col];
      }
    }
  }
  floatx16 acc[4];
  bf16x8 Bq[MAXT][2];
  __syncthreads();  // P0
  __syncthreads();  // P
  Tile cur = read_slot(0);
  const __amdgpu_buffer_rsrc_t wrs =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (int)(4u * (unsigned)cpb * chunk_b), 0x00020000);
  // slot i of the ring holds tap t = 2*i + kh of a chunk
  auto fetch_chunk_tap = [&](int chunk, int i) {
    const unsigned so = (unsigned)chunk * chunk_b + (unsigned)(2 * i + kh) * tap_b;
    Bq[i][0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, ulane, so, 0));
    Bq[i][1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, ulane, so + plane_b, 0));
  };
#pragma unroll
  for (int i = 0; i < MAXT; ++i)
    if (2 * i + kh < 9) fetch_chunk_tap(0, i);   // chunk 0 is block 0: 9 taps
  int s_nxt = 1;
  while (cur.item < a.ntiles) {
    const Tile nxt = read_slot(s_nxt);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
      const int blk = chunk / cpb;                                // wave-uniform
      const int nv = 3 - (blk & 1), ntaps = (3 - (blk >> 1)) * nv;
      const int nchunk = chunk + 1 < nchunks ? chunk + 1 : 0;     // the ring runs on into the next tile
      const int nblk = nchunk / cpb;
      const int ntaps_n = (3 - (nblk >> 1)) * (3 - (nblk & 1));
      const unsigned char* wbuf = lds_raw + (chunk & 1) * WINB + apix;
#pragma unroll
      for (int i = 0; i < MAXT; ++i) {
        const int t = 2 * i + kh;
        if (t < ntaps) {                                          // wave-uniform
          const int u = t / nv, v = t - u * nv;                   // window offset of the tap: (u, v) (halo 1 included)
          const unsigned char* ap = wbuf + u * ROWB + v * REC;
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const bf16x8 Ah = *(const bf16x8*)(ap + m * 2 * ROWB);
            const bf16x8 Am = *(const bf16x8*)(ap + m * 2 * ROWB + 32);
            floatx16 c = acc[m];
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bq[i][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bq[i][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bq[i][0], c, 0, 0, 0);
            acc[m] = c;
          }
        }
        if (t < ntaps_n) fetch_chunk_tap(nchunk, i);              // same slot, next chunk
      }
      __syncthreads();  // B_chunk
    }

    // ---- reduction over the two tap halves, bias + activation, into the output region ---------
    {
      float* yb = yreg + (4 * h) * YP + nt * 32 + l31;   // + (m*32 + (e&3) + 8*(e>>2)) * YP
      if (kh == 1) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int e = 0; e < 16; ++e) yb[(m * 32 + (e & 3) + 8 * (e >> 2)) * YP] = acc[m][e];
      }
      __syncthreads();  // E1
      if (kh == 0) {
        auto finish = [&](auto act_tag) {
          constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
          for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
              float* p0 = yb + (m * 32 + (e & 3) + 8 * (e >> 2)) * YP;
              float* p1 = yb + (m * 32 + ((e + 1) & 3) + 8 * ((e + 1) >> 2)) * YP;
              floatx2 v = {acc[m][e] + *p0, acc[m][e + 1] + *p1};
              v = v + floatx2{pbias, pbias};
              if (ACT == DSIC_ACT_GDN) {
                v = gdn_pair<false>(v, floatx2{pbeta, pbeta}, floatx2{pgamma, pgamma});
              } else if (ACT == DSIC_ACT_RELU) {
                v[0] = v[0] > 0.f ? v[0] : 0.f;
                v[1] = v[1] > 0.f ? v[1] : 0.f;
              }
              *p0 = v[0];
              *p1 = v[1];
            }
          }
        };
        if (a.act == DSIC_ACT_GDN)
          finish(std::integral_constant<int, DSIC_ACT_GDN>{});
        else if (a.act == DSIC_ACT_RELU)
          finish(std::integral_constant<int, DSIC_ACT_RELU>{});
        else
          finish(std::integral_constant<int, DSIC_ACT_NONE>{});
      }
    }
    __syncthreads();  // E2
    cur = nxt;
    s_nxt = s_nxt == 2 ? 0 : s_nxt + 1;
  }
}

// w [Cout][Cs][5][5] fp32 -> bf16 planes [blk 4][Cs/16][tap 9][plane 2][CoutP][16]; tap t of block (a,b):
// (u, v) = (t / nv, t % nv), ky = 2u + a, kx = 2v + b; slots beyond a block's taps stay zero (never read)
__global__ void pack5s2_kernel(const float* __restrict__ w, unsigned short* __restrict__ dst, int Cout, int Cs,
                               int CoutP, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over [blk][cc][tap][n][k]
  if (i >= total) return;
  const int k = i & 15;
  int64_t r = i >> 4;
  const int n = r % CoutP;
  r /= CoutP;
  const int t = r % 9;
  r /= 9;
  const int cpb = Cs / CK;
  const int cc = r % cpb;
  const int blk = r / cpb;
  const int pa = blk >> 1, pb = blk & 1;
  const int nv = 3 - pb, ntaps = (3 - pa) * nv;
  float v = 0.f;
  if (t < ntaps && n < Cout) {
    const int u = t / nv, vv = t - u * nv;
    const int ky = 2 * u + pa, kx = 2 * vv + pb;
    v = w[(((size_t)n * Cs + cc * CK + k) * 5 + ky) * 5 + kx];
  }
  const size_t base = ((((size_t)blk * cpb + cc) * 9 + t) * 2) * CoutP;
  const __bf16 hi = (__bf16)v;
  const float rem = v - (float)hi;
  const __bf16 mid = (__bf16)rem;
  dst[(base + n) * 16 + k] = __builtin_bit_cast(unsigned short, hi);
  dst[(base + CoutP + n) * 16 + k] = __builtin_bit_cast(unsigned short, mid);
}

}  // namespace c5
}  // namespace dsic

using namespace dsic;

extern "C" int64_t dsic_conv5s2_bf16_weight_bytes(int Cout, int Cs) {
  return (int64_t)4 * (Cs / c5::CK) * 9 * 2 * round_up(Cout, 32) * 16 * 2;
}

extern "C" int dsic_pack_conv5s2_bf16_weight(const float* w_oihw5, void* dst, int Cout, int Cs, void* stream) {
  DSIC_REQUIRE(w_oihw5 && dst && Cout > 0 && Cs >= 64 && Cs % 16 == 0, "pack_conv5s2_bf16_weight: bad argument");
  const int CoutP = round_up(Cout, 32);
  const int64_t total = (int64_t)4 * (Cs / c5::CK) * 9 * CoutP * 16;
  hipLaunchKernelGGL(c5::pack5s2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     w_oihw5, (unsigned short*)dst, Cout, Cs, CoutP, total);
  return check_launch("pack_conv5s2_bf16_weight");
}

extern "C" int dsic_conv5s2_bf16_nhwc(const float* in_s2d, const void* w_planes, const float* bias, const float* beta,
                                      const float* gamma, float* out, int B, int H, int W, int Cs, int Cout, int act,
                                      void* ticket, void* stream) {
  DSIC_REQUIRE(in_s2d && w_planes && bias && out && ticket, "conv5s2_bf16: null pointer");
  DSIC_REQUIRE(B > 0 && H > 0 && W > 0, "conv5s2_bf16: empty tensor");
  DSIC_REQUIRE(Cs >= 64 && Cs % 16 == 0, "conv5s2_bf16: Cs=%d must be a multiple of 16, >= 64", Cs);
  DSIC_REQUIRE(Cout > 0 && Cout % 4 == 0 && Cout <= 128, "conv5s2_bf16: Cout=%d must be a multiple of 4, <= 128", Cout);
  DSIC_REQUIRE(act == DSIC_ACT_NONE || act == DSIC_ACT_GDN || act == DSIC_ACT_RELU, "conv5s2_bf16: act=%d", act);
  DSIC_REQUIRE(act != DSIC_ACT_GDN || (beta && gamma), "conv5s2_bf16: GDN needs beta and gamma");
  c5::Args a{};
  a.in = in_s2d; a.w = w_planes; a.bias = bias; a.beta = beta; a.gamma = gamma; a.out = out;
  a.B = B; a.H = H; a.W = W; a.Cs = Cs; a.Cout = Cout; a.CoutP = round_up(Cout, 32); a.act = act;
  a.ticket = (unsigned long long*)ticket;
  a.tiles_x = ceil_div(W, 16);
  a.tiles_y = ceil_div(H, 8);
  const int64_t nt = (int64_t)a.tiles_x * a.tiles_y * B;
  DSIC_REQUIRE(nt < ((int64_t)1 << 31), "conv5s2_bf16: too many tiles");
  DSIC_REQUIRE((int64_t)H * W * 4 * Cs * 4 < ((int64_t)1 << 31) && (int64_t)H * W * Cout * 4 < ((int64_t)1 << 31),
               "conv5s2_bf16: one image must stay below 2 GiB (32-bit offsets inside an image)");
  a.ntiles = (int)nt;
  a.nt_out = (int64_t)B * H * W * Cout * 4 > (300ll << 20);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  static bool attr_set[64] = {};
  if (!attr_set[dev]) {
    const void* fns[2] = {(const void*)c5::conv5s2_bf16_kernel<false>, (const void*)c5::conv5s2_bf16_kernel<true>};
    for (int i = 0; i < 2; ++i) {
      const hipError_t e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, c5::LDS_TOTAL);
      if (e != hipSuccess) {
        set_error("conv5s2_bf16: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return DSIC_EHIP;
      }
    }
    attr_set[dev] = true;
  }
  static int max_grid = 0;
  if (max_grid == 0) {
    const char* g = getenv("DSIC_WINO_GRID");
    max_grid = g ? atoi(g) : 256;
    if (max_grid < 1 || max_grid > 1024) max_grid = 256;
  }
  const int grid = a.ntiles < max_grid ? a.ntiles : max_grid;
  if (a.nt_out)
    hipLaunchKernelGGL(c5::conv5s2_bf16_kernel<true>, dim3(grid), dim3(c5::THREADS), c5::LDS_TOTAL, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(c5::conv5s2_bf16_kernel<false>, dim3(grid), dim3(c5::THREADS), c5::LDS_TOTAL, (hipStream_t)stream, a);
  return check_launch("conv5s2_bf16");
}
