// Last synthesis layer: ConvTranspose2d(Cin, Cimg, 5, stride 2, padding 2, output_padding 1) from NHWC
// features to the NCHW image (code/modelv2/layers.py:98, "deconv(N, out_ch)").
//
// With only 3 (or 4) output channels the layer is a skinny GEMM: N = 4 sub-pixel phases x Cimg
// <= 16 columns, K = 9 taps x Cin.  Output pixel (2i+py, 2j+px) reads input (i-1+wr, j-1+wc) with
//   g[wr][wc] = w[ci][c][py+4-2wr][px+4-2wc]  for wr >= py, wc >= px, else 0,
// i.e. one 3x3 stride-1 convolution over the input grid whose columns are (phase, c).  It runs on
// v_mfma_f32_16x16x4_f32 (N = 16: 12 of 16 columns useful for RGB) instead of the 32-column tiles of
// conv_igemm.hip (12 of 32 useful): half the MFMA work for the same result, same exact-fp32 products.
//
// Workgroup = 256 threads = 4 waves on a 32x16-pixel tile of the input grid (64x32 output pixels).
// Per 16-channel chunk the 34x18 window is staged in LDS (pixel stride 20 floats: conflict-free
// ds_read_b128), next chunk's global loads in flight during the MFMAs.  Wave w owns rows 4w..4w+3:
// 8 M tiles of 16 pixels.  MFMA operand map (lane l: r = l&15, q = l>>4): A[i=r][k=q], B[k=q][j=r],
// D[i=4q+v][j=r]; k-step s of a chunk pairs channels {4q+s}, so both operands are one 16-byte read
// per lane for four MFMAs.  Weights are packed [tap][Cin/16][col 16][16 ch] (73.7 KB for Cin=128,
// L1/L2 resident).  Epilogue: + bias, interleave the phases through LDS, 16-byte NCHW stores.
#include <stdlib.h>

#include "common.h"

namespace dsic {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef short shortx4 __attribute__((ext_vector_type(4)));
typedef unsigned uintx2 __attribute__((ext_vector_type(2)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// fp32 quad -> two bf16 planes (hi = bf16(v), mid = bf16(v - hi)), each 4 bf16 = 8 bytes.  The conversions
// are compiler builtins, not inline asm: their results feed MFMA operands directly, and only for
// instructions it knows does the compiler insert the wait states a VALU write needs before an MFMA read.
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void img_split(floatx4 v, uintx2& hi, uintx2& mid) {
  hi = __builtin_bit_cast(uintx2, __builtin_convertvector(v, bf16x4));
  floatx4 r;
  r[0] = v[0] - __builtin_bit_cast(float, hi[0] << 16);
  r[1] = v[1] - __builtin_bit_cast(float, hi[0] & 0xFFFF0000u);
  r[2] = v[2] - __builtin_bit_cast(float, hi[1] << 16);
  r[3] = v[3] - __builtin_bit_cast(float, hi[1] & 0xFFFF0000u);
  mid = __builtin_bit_cast(uintx2, __builtin_convertvector(r, bf16x4));
}

constexpr int IT_W = 32, IT_H = 16;          // input-grid pixels per workgroup tile
constexpr int IW = IT_W + 2, IH = IT_H + 2;  // staged window
constexpr int ICK = 16, IP = ICK + 4;        // channels per chunk, LDS floats per window pixel
constexpr int ISLOTS = (IH * IW * (ICK / 4) + 255) / 256;  // float4 staging slots per thread

struct ImgArgs {
  const float* in;
  const float* w;  // packed [9][Cin/16][16][16]
  const float* bias;
  float* out;  // [B][Cimg][2H][2W]
  int B, H, W, Cin, Cimg;
  int tiles_x, tiles_y;
  int ntiles;      // convT_image_dma_kernel: tiles_x * tiles_y * B (tiles of 16x16 input pixels)
};

// timing-only ablations (wrong results): 1 no global window loads, 2 no MFMAs, 4 no weight loads / splits,
// 8 no output stores, 16 no window split + LDS stores, 32 no LDS A reads
#ifndef IMG_ABL
#define IMG_ABL 0
#endif
// -DIMG_STAMP=1 (diagnostic build, destroys the first output row): waves 0 and 4 of workgroup 0 write the
// s_memtime stamps of their phases 16..31 (third and fourth tile) over out[0..127]; tools/img_stamps.py prints them
#ifndef IMG_STAMP
#define IMG_STAMP 0
#endif
#if IMG_STAMP
#define ISTAMP(i) do { if (phi >= 16 && phi < 32) st[(phi - 16) * 4 + (i)] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define ISTAMP(i) do { } while (0)
#endif
#ifndef IMG_WGS
#define IMG_WGS 2  // workgroups per CU the register budget is set for (3 spills and is slower)
#endif
// BF16 = true: the contraction runs on v_mfma_f32_16x16x16_bf16 with both operands split into two
// bf16 planes (products hi*hi, hi*mid, mid*hi, fp32 accumulate - the scheme of conv_wino_bf16.hip):
// three bf16 MFMAs replace four fp32-input ones at a quarter of their cycles each.  The window is
// split once while it is staged (LDS pixel record: 16 bf16 hi | 16 bf16 mid | pad = the same 80
// bytes as 16 fp32 + pad), the weight quads when they are fetched.
template <bool BF16>
__global__ __launch_bounds__(256, IMG_WGS) void convT_image_kernel(const ImgArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[IH * IW * IP];  // 48 960 B; reused as the output tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  int bx = blockIdx.x;
  const int tile_x = bx % a.tiles_x;
  bx /= a.tiles_x;
  const int tile_y = bx % a.tiles_y;
  const int n = bx / a.tiles_y;
  const int x0 = tile_x * IT_W, y0 = tile_y * IT_H;
  const int Cin = a.Cin, C16 = Cin >> 4;

  // staging slots of this thread: window pixel + channel quad -> global float offset (or -1) and LDS offset
  const float* img = a.in + (size_t)n * a.H * a.W * Cin;
  int goff[ISLOTS], loff[ISLOTS];
#pragma unroll
  for (int i = 0; i < ISLOTS; ++i) {
    const int slot = tid + i * 256;
    int g = -1, lo = -1;
    if (slot < IH * IW * (ICK / 4)) {
      const int pix = slot >> 2, cq = slot & 3;
      const int wy = pix / IW, wx = pix % IW;
      const int gy = y0 - 1 + wy, gx = x0 - 1 + wx;
      lo = pix * IP + cq * 4;
      if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) g = (gy * a.W + gx) * Cin + cq * 4;
    }
    goff[i] = g;
    loff[i] = lo;
  }
  // Two chunks (32 channels = one 128-byte line per pixel) are loaded together and staged one after the other:
  // loaded a chunk apart, the second 64-byte half of every line missed L2 again (2.5 MB of windows in flight per
  // XCD pull 5 MB of lines through a 4 MB L2): 7.75 M 128-byte memory-side read requests per launch = 991 MB for
  // a 537 MB input (TCC_EA0_RDREQ).
  floatx4 stage[2][ISLOTS];
  auto load_pair = [&](int chunk) {   // chunks `chunk` (even) and chunk + 1
#pragma unroll
    for (int i = 0; i < ISLOTS; ++i) {
      const floatx4 z = {0.f, 0.f, 0.f, 0.f};
      if (IMG_ABL & 1) { stage[0][i] = z; stage[1][i] = z; continue; }
      stage[0][i] = goff[i] >= 0 ? *(const floatx4*)(img + goff[i] + chunk * ICK) : z;
      stage[1][i] = goff[i] >= 0 && chunk + 1 < C16 ? *(const floatx4*)(img + goff[i] + (chunk + 1) * ICK) : z;
    }
  };
  auto store_chunk = [&](int half) {
    if (IMG_ABL & 16) return;
#pragma unroll
    for (int i = 0; i < ISLOTS; ++i)
      if (loff[i] >= 0) {
        const floatx4 sv = half ? stage[1][i] : stage[0][i];
        if (BF16) {
          // pixel record: hi quads at +0..31, mid quads at +32..63 (bytes); loff = pix*IP + cq*4 floats
          uintx2 hi, mid;
          img_split(sv, hi, mid);
          unsigned char* rec = (unsigned char*)lds + (size_t)(loff[i] / IP) * (IP * 4);
          const int cq = (loff[i] % IP) >> 2;
          *(uintx2*)(rec + cq * 8) = hi;
          *(uintx2*)(rec + 32 + cq * 8) = mid;
        } else {
          *(floatx4*)(lds + loff[i]) = sv;
        }
      }
  };

  floatx4 acc[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) acc[m] = floatx4{0.f, 0.f, 0.f, 0.f};
  // A fragment of M tile m (row m>>1 of this wave, x half m&1) at tap (0,0): + (wr*IW + wc)*IP per tap
  const int abase = ((wave * 4) * IW + r) * IP + 4 * q;
  const float* wl = a.w + r * 16 + 4 * q;  // + (tap*C16 + chunk)*256

  load_pair(0);
  for (int chunk = 0; chunk < C16; ++chunk) {
    __syncthreads();  // the previous chunk's window is no longer read
    store_chunk(chunk & 1);
    __syncthreads();
    if ((chunk & 1) && chunk + 1 < C16) load_pair(chunk + 1);  // in flight during the MFMAs below
    floatx4 b = *(const floatx4*)(wl + (size_t)(0 * C16 + chunk) * 256);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int wr = tap / 3, wc = tap % 3;
      const floatx4 bc = b;
      if (tap + 1 < 9 && !(IMG_ABL & 4)) b = *(const floatx4*)(wl + (size_t)((tap + 1) * C16 + chunk) * 256);
      if (BF16) {
        uintx2 bh, bm;
        if (IMG_ABL & 4) { bh = uintx2{(unsigned)tid, 1u}; bm = uintx2{2u, (unsigned)lane}; }
        else img_split(bc, bh, bm);
        const shortx4 Bh = __builtin_bit_cast(shortx4, bh), Bm = __builtin_bit_cast(shortx4, bm);
        // A fragment of lane (r, q): channels 4q..4q+3 of pixel r: 8 bytes of the hi part, 8 of the mid part
        const unsigned char* abyte = (const unsigned char*)lds + (size_t)((wave * 4) * IW + r) * (IP * 4) + q * 8;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          const unsigned char* ap = abyte + (size_t)((((m >> 1) + wr) * IW + (m & 1) * 16 + wc) * (IP * 4));
          shortx4 Ah, Am;
          if (IMG_ABL & 32) { Ah = shortx4{(short)m, (short)tap, 1, 2}; Am = Ah; }
          else { Ah = *(const shortx4*)ap; Am = *(const shortx4*)(ap + 32); }
          if (IMG_ABL & 2) {
            acc[m][0] += (float)(Ah[0] + Am[1] + Bh[2] + Bm[3]);
          } else {
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(Ah, Bm, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(Am, Bh, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(Ah, Bh, acc[m], 0, 0, 0);
          }
        }
      } else {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          const floatx4 av = *(const floatx4*)(lds + abase + (((m >> 1) + wr) * IW + (m & 1) * 16 + wc) * IP);
#pragma unroll
          for (int s = 0; s < 4; ++s) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bc[s], acc[m], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue: (phase, c) columns -> interleaved output pixels, through LDS --------------------
  __syncthreads();  // window dead
  const int Cimg = a.Cimg;
  const int phase = r / Cimg, c = r - phase * Cimg;  // column r = phase*Cimg + c, valid if r < 4*Cimg
  if (r < 4 * Cimg) {
    const float bias = a.bias[c];
    const int py = phase >> 1, px = phase & 1;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int yl = wave * 4 + (m >> 1);
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int xl = (m & 1) * 16 + 4 * q + v;
        lds[(c * (2 * IT_H) + 2 * yl + py) * (2 * IT_W) + 2 * xl + px] = __fadd_rn(acc[m][v], bias);
      }
    }
  }
  __syncthreads();
  const int OH = 2 * a.H, OW = 2 * a.W;
  const int oy0 = 2 * y0, ox0 = 2 * x0;
  const bool vec = (OW & 3) == 0;
  for (int idx = tid; idx < Cimg * (2 * IT_H) * (2 * IT_W / 4); idx += 256) {
    const int x4 = idx % (2 * IT_W / 4);
    const int oy = (idx / (2 * IT_W / 4)) % (2 * IT_H);
    const int cc = idx / ((2 * IT_W / 4) * (2 * IT_H));
    const int gy = oy0 + oy, gx = ox0 + 4 * x4;
    if (gy >= OH || gx >= OW) continue;
    const floatx4 v = *(const floatx4*)(lds + (cc * (2 * IT_H) + oy) * (2 * IT_W) + 4 * x4);
    float* dst = a.out + (((size_t)n * Cimg + cc) * OH + gy) * OW + gx;
    if (IMG_ABL & 8) {
      if (v[0] == 1.2345e-30f) dst[0] = v[1];
    } else if (vec && gx + 3 < OW) {
      *(floatx4*)dst = v;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (gx + e < OW) dst[e] = v[e];
    }
  }
}


// ================================================================================================================
// convT_image_dma_kernel (round 3): the same layer for Cin <= 128, built so that nothing in the contraction loop
// touches global memory.  (Ablations of the kernel above, -DIMG_ABL: of its 0.29 ms, 0.14 ms are the window loads -
// they sit in 80 staging registers, one chunk pair at a time, and the weight loads of the MFMA loop queue behind
// them in the in-order vector-memory path - 0.08 ms MFMAs, 0.05 ms weight loads + splits.)
//   * Persistent workgroup of 512 threads (8 waves) per CU over tiles of 16x16 input pixels (32x32 output pixels).
//   * All weights live in LDS as bf16 planes (pack kernel's second half, copied once per workgroup): 9216 B per
//     16-channel chunk, [tap 9][plane 2][half 2][column 16][8 channels].
//   * The 18x18 window of a 16-channel chunk (fp32, 20.7 KB) goes global -> LDS by buffer_load_dwordx4 ... lds into
//     a ring of four chunk buffers, two chunks (= one 128-byte line per pixel) per request and two requests ahead
//     (40 KB in flight per CU).  A pixel's 64 bytes stay contiguous (four lanes of a DMA instruction read 64
//     contiguous bytes: with one pixel per lane, 16 bytes out of every line and instruction, the kernel took as long
//     as the staged one); its four 16-byte pieces (j, half) = channels 8 half + 4 j .. + 3 are XOR-swizzled with
//     bits 2..3 of the pixel number, so that 16 consecutive pixels of one piece hit 16 different bank groups:
//     conflict-free ds_read_b128 without padding.
//   * In-place split: thread (pixel, half) reads its two fp32 pieces, writes the 8 hi parts over the first and the 8
//     mid parts over the second - piece (plane, half) of a pixel then is 8 bf16, which is the A operand of
//     v_mfma_f32_16x16x32_bf16 for lane (pixel r, k-block q): plane q >> 1, half q & 1, i.e. K = 32 holds
//     [hi 16 channels | mid 16 channels] of ONE 16-channel chunk.  With B = [Uhi | Uhi] and then [Umid | Umid] two
//     MFMAs give (Vhi + Vmid)(Uhi + Umid): all four products of the two-plane split (one more than the
//     three-product scheme of the Winograd kernels) in 32 cycles instead of 3 x 16 for the legacy 16x16x16 form.
//   * Waves 0..3 contract the even chunk of a pair, waves 4..7 the odd one (wave & 3 = four tile rows = 4 M tiles;
//     an A fragment of a window row serves up to three (row, tap row) combinations; the chunk's 18 B fragments are
//     held in registers): per chunk and wave 18 + 18 ds_read_b128 for 72 MFMAs.  The two partial sums are added
//     in fixed order (even + odd, then + bias) in the output tile in LDS, which leaves as 16-byte NCHW stores.
namespace imgd {
constexpr int TW = 16, TH = 16, WW = TW + 2, WH = TH + 2, NPIX = WW * WH;  // 324 window pixels
constexpr int CB = NPIX * 64;        // chunk buffer: 20 736 B
constexpr int THREADS = 512;
// weight bytes per 16-channel chunk in LDS: [tap 9][plane 2][half 2][column NCOL][8 channels]; with NCOL = 12 (up to
// three image channels) the eight chunks of Cin = 128 leave room for a fifth window buffer
__host__ __device__ constexpr int wch(int ncol) { return 9 * 2 * 2 * ncol * 16; }
constexpr int NKI = (NPIX + 15) / 16;          // 21 DMA instructions of 16 pixels per chunk (the last one: 4 pixels)
constexpr int NKW = (NKI + 3) / 4;             // at most 6 of them per wave (wave wv of the splitting group: wv, wv + 4, ...)
constexpr int MAXCH = 8;             // Cin <= 128
inline int lds_bytes(int Cin, int nbuf, int ncol) { return nbuf * CB + (Cin / 16) * wch(ncol); }
// byte offset of piece c (0..3) of window pixel p inside a chunk buffer: the pixel's 64 bytes are contiguous (one
// DMA lane quad = 64 contiguous bytes of global memory), the piece slot is XOR-swizzled with bits 2..3 of the
// pixel number so that 16 consecutive pixels of one piece cover all 16 16-byte bank groups (conflict-free b128)
__device__ __forceinline__ int piece_off(int p, int c) { return p * 64 + ((c ^ ((p >> 2) & 3)) << 4); }
}  // namespace imgd

// NBUF = 5, NCOL = 12: Cimg <= 3; NBUF = 4, NCOL = 16: Cimg = 4.  Chunk phi + NBUF - 2 is requested in phase phi.
template <int NBUF, int NCOL>
__global__ __launch_bounds__(imgd::THREADS) void convT_image_dma_kernel(const ImgArgs a) {
  using namespace imgd;
  constexpr int WCH = wch(NCOL);
  constexpr int LEAD = NBUF - 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  unsigned char* const ring = lds_raw;
  unsigned char* const wl = lds_raw + NBUF * CB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int grp = wave >> 2, wv = wave & 3;
  const int Cin = a.Cin, nch = Cin >> 4;   // even (host)
  // XCD-aware order: workgroup b runs on XCD b % 8; in every round the 32 workgroups of an XCD take 32 consecutive
  // tiles (half of a 128x128 image), so the halo lines of neighbouring tiles meet in one L2
  const int nwg = gridDim.x;
  const int per_xcd = nwg >> 3;
  const int vid = (nwg & 7) ? (int)blockIdx.x : ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
  const int my_tiles = vid < a.ntiles ? (a.ntiles - vid + nwg - 1) / nwg : 0;
  const int S = my_tiles * nch;            // chunks of this workgroup, in order

  {  // weights -> LDS (bf16 planes: the second half of the packed buffer, 16 columns per block there)
    const unsigned char* wg = (const unsigned char*)(a.w + (size_t)9 * Cin * 16);
    for (int i = tid; i < nch * 36 * NCOL; i += THREADS) {
      const int blk = i / NCOL, col = i - blk * NCOL;
      *(uintx4*)(wl + i * 16) = *(const uintx4*)(wg + (blk * 16 + col) * 16);
    }
  }

  // ---- window requests -------------------------------------------------------------------------------------
  // DMA instruction ki (16 pixels): lane -> pixel 16 ki + (lane >> 2), slot lane & 3 of the pixel's 64 bytes in LDS
  // (the instruction fills 1 KB at M0 + 16 lane); the slot holds piece slot ^ swizzle(pixel) = (j, half), whose
  // raw content is the channel quad 2 half + j of the chunk.  The four waves of a group issue the requests of the
  // chunks of their own parity: wave wv the instructions ki = wv, wv + 4, ... < NKI.
  int wyx[NKW];   // window row << 8 | window column, or -1; bits 20.. : the channel quad
#pragma unroll
  for (int k = 0; k < NKW; ++k) {
    const int pix = 16 * (wv + 4 * k) + (lane >> 2);
    const int wy = pix / WW;
    const int c = (lane & 3) ^ ((pix >> 2) & 3);
    const int qd = 2 * (c & 1) + (c >> 1);
    wyx[k] = (wv + 4 * k < NKI && pix < NPIX) ? (qd << 20) | (wy << 8) | (pix - wy * WW) : -1;
  }
  const int ndma = wv + 4 * (NKW - 1) < NKI ? NKW : NKW - 1;   // loads of this wave per chunk of its parity: 6 or 5
  unsigned off[NKW];
  uintx4 rsrc = {0u, 0u, 0u, 0x00020000u};
  const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)lds_raw;
  auto tile_of = [&](int ti, int& n, int& ty, int& tx) {
    int t = vid + ti * nwg;
    tx = t % a.tiles_x;
    t /= a.tiles_x;
    ty = t % a.tiles_y;
    n = t / a.tiles_y;
  };
  auto aim = [&](int ti) {   // per tile: the lanes' byte offsets into the image, the image's buffer descriptor
    int n, ty, tx;
    tile_of(ti, n, ty, tx);
    const unsigned long long base = (unsigned long long)(a.in + (size_t)n * a.H * a.W * Cin);
    rsrc[0] = (unsigned)base;
    rsrc[1] = (unsigned)(base >> 32) & 0xFFFFu;
    rsrc[2] = (unsigned)(a.H * a.W * Cin * 4);
#pragma unroll
    for (int k = 0; k < NKW; ++k) {
      const int gy = ty * TH - 1 + ((wyx[k] >> 8) & 255), gx = tx * TW - 1 + (wyx[k] & 255);
      const bool ok = wyx[k] >= 0 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && !(IMG_ABL & 1);
      off[k] = ok ? (unsigned)(((gy * a.W + gx) * Cin) * 4 + (wyx[k] >> 20) * 16) : 0x80000000u;   // out of range: zeros
    }
  };
  // requests of chunk s = tile ti, chunk c of the tile (the caller aims first)
  auto dma = [&](int s, int c) {
    const uintx4 rs = {(unsigned)__builtin_amdgcn_readfirstlane((int)rsrc[0]), (unsigned)__builtin_amdgcn_readfirstlane((int)rsrc[1]),
                       (unsigned)__builtin_amdgcn_readfirstlane((int)rsrc[2]), (unsigned)__builtin_amdgcn_readfirstlane((int)rsrc[3])};
    const unsigned soff = (unsigned)(c * 64);
    const unsigned dst = lds_base + (unsigned)((s % NBUF) * CB);
#pragma unroll
    for (int k = 0; k < NKW; ++k) {
      const int ki = wv + 4 * k;
      if (ki >= NKI) break;
      const unsigned m0v = dst + (unsigned)(ki * 1024);
      if (ki < NKI - 1 || lane < 4 * (NPIX - 16 * (NKI - 1)))
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                     :: "s"(m0v), "v"(off[k]), "s"(rs), "s"(soff) : "memory");
    }
  };
  // all vector-memory operations of this wave but the newest `newer` ones are complete
  auto wait_older = [&](int newer) {
    if (newer == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (newer == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  };

  // A fragment offsets inside a chunk buffer: window rows 4 wv + rho (rho < 6), columns wc + r; piece q
  int aoff[18];
#pragma unroll
  for (int rho = 0; rho < 6; ++rho)
#pragma unroll
    for (int wc = 0; wc < 3; ++wc) aoff[rho * 3 + wc] = piece_off((4 * wv + rho) * WW + wc + r, q);

  // output side: column r = (phase, c); the lane of the other x phase; this lane's four output columns of an M tile
  const int Cimg = a.Cimg;
  const int phase = r / Cimg, oc = r - phase * Cimg;
  const bool colok = r < 4 * Cimg;
  const int py = phase >> 1, px = phase & 1;
  const int partner4 = (colok ? (px ? lane - Cimg : lane + Cimg) : lane) * 4;
  const float bias = colok ? a.bias[oc] : 0.f;
  const int OH = 2 * a.H, OW = 2 * a.W;
  // the 16 pixels x 2 x phases of M tile y = 32 consecutive floats of output row 2 (4 wv + y) + py: this lane holds
  // columns 8 q + 4 px .. + 3 after trading two values with the lane of the other x phase
  auto gather = [&](const floatx4 v) {
    const float s0 = px ? v[0] : v[2], s1 = px ? v[1] : v[3];
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(partner4, __builtin_bit_cast(int, s0)));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(partner4, __builtin_bit_cast(int, s1)));
    return px ? floatx4{r0, v[2], r1, v[3]} : floatx4{v[0], r0, v[1], r1};
  };
  struct OutPos { float* p; int gy, gx; };
  auto out_pos = [&](int ti, int y) {
    int n, ty, tx;
    tile_of(ti, n, ty, tx);
    OutPos o;
    o.gy = 2 * (ty * TH + 4 * wv + y) + py;
    o.gx = 2 * tx * TW + 8 * q + 4 * px;
    o.p = a.out + (((size_t)n * Cimg + oc) * OH + o.gy) * OW + o.gx;
    return o;
  };
  const bool vec = (OW & 3) == 0;

  __syncthreads();   // weights in LDS
  int aimed = 0;
  if (S > 0) {
    aim(0);
    dma(grp, grp);   // chunks 0 and 1 (nch >= 2), each by the group of its parity
    wait_older(0);
  }
  __syncthreads();   // chunks 0 and 1 have landed
  if (LEAD == 3 && grp == 0 && 2 < S) {   // chunk 2: its buffer is fresh
    if (2 / nch != aimed) { aimed = 2 / nch; aim(aimed); }
    dma(2, 2 % nch);
  }

  floatx4 acc[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) acc[m] = floatx4{0.f, 0.f, 0.f, 0.f};
  // Phase phi (between two barriers): the waves of parity phi & 1 split chunk phi in place and fetch the B fragments
  // of that chunk, which they contract in phase phi + 1 while the other group splits (vector instructions of the
  // two waves of a SIMD do not overlap, but the splitting wave's LDS latencies hide behind the other wave's MFMAs).
  // The splitting group also requests chunk phi + LEAD (its buffer was read last in phase phi - 1) before it splits,
  // behind its output stores, and makes sure that its earlier requests have landed: LEAD = 3: at the end of the
  // phase, all but the newest (chunk phi + 1, requested two phases ago); LEAD = 2: at the end of the next phase,
  // where it contracts and requests nothing.  A contracting wave with LEAD = 3 never waits for memory.
  // A tile's last even chunk is contracted in phase (t+1) nch - 1, its last odd one a phase later: group 0 stores
  // bias + its sum at the start of phase (t+1) nch, group 1 adds its sum to that a phase later.
  int pc = 0, pt = 0;            // phi = pt * nch + pc
  int dc = LEAD % nch, dt = LEAD / nch;   // chunk phi + LEAD
  bf16x8 Bf[9][2];
#if IMG_STAMP
  unsigned st[64];
  for (int i = 0; i < 64; ++i) st[i] = 0;
#endif
  for (int phi = 0; phi <= S + 1; ++phi) {
    ISTAMP(0);
    const bool splitter = (phi & 1) == grp;
#ifndef IMG_NOPRIO
    // the splitting wave's few vector instructions go first: left to the default arbitration they wait until the
    // contracting wave of their SIMD has issued its last MFMA, and the phase lasts MFMA time + split time
    if (splitter) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
#endif
    const bool fin0 = grp == 0 && pc == 0 && pt > 0 && phi <= S;          // tile pt - 1, this wave's sum complete
    const bool fin1 = grp == 1 && pc == 1 && pt > 0 && phi <= S + 1;      // tile pt - 1 (nch >= 2: pc == 1 exists)
    floatx4 prev[4];
    if (fin0 && colok && !(IMG_ABL & 8)) {
#pragma unroll
      for (int y = 0; y < 4; ++y) {
        floatx4 v = acc[y];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = __fadd_rn(v[e], bias);
        v = gather(v);
        const OutPos o = out_pos(pt - 1, y);
        if (o.gy < OH) {
          if (vec && o.gx + 3 < OW) *(floatx4*)o.p = v;
          else
            for (int e = 0; e < 4; ++e)
              if (o.gx + e < OW) o.p[e] = v[e];
        }
      }
    }
    if (fin0) {
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[m] = floatx4{0.f, 0.f, 0.f, 0.f};
    }
    if (fin1 && colok && !(IMG_ABL & 8)) {
#pragma unroll
      for (int y = 0; y < 4; ++y) {
        const OutPos o = out_pos(pt - 1, y);
        prev[y] = floatx4{0.f, 0.f, 0.f, 0.f};
        if (o.gy < OH) {
          if (vec && o.gx + 3 < OW) prev[y] = __builtin_nontemporal_load((const floatx4*)o.p);
          else
            for (int e = 0; e < 4; ++e)
              if (o.gx + e < OW) prev[y][e] = __builtin_nontemporal_load(o.p + e);
        }
      }
    }
    int newer = 0;
    if (splitter && phi + LEAD < S && !(IMG_ABL & 64)) {
      if (dt != aimed) { aimed = dt; aim(dt); }
      dma(phi + LEAD, dc);
      newer = ndma;
    }
    ISTAMP(1);
    if (splitter && phi < S && !(IMG_ABL & 16)) {
      // ---- in-place split of chunk phi: 2 NPIX items (half, pixel) over the 256 threads of the group ---------
      unsigned char* const bp = ring + (phi % NBUF) * CB;
      const int t4 = tid & 255;
      floatx4 x0[3], x1[3];
      unsigned char *p0[3], *p1[3];
#pragma unroll
      for (int sl = 0; sl < 3; ++sl) {
        int item = t4 + sl * 256;
        if (item >= 2 * NPIX) item = 2 * NPIX - 1;   // the idle lanes of the last slot redo item 647 (stores masked)
        const int half = item >= NPIX ? 1 : 0;
        const int pix = item - half * NPIX;
        p0[sl] = bp + piece_off(pix, half);        // raw quad 2 half     -> hi plane of this half
        p1[sl] = bp + piece_off(pix, 2 + half);    // raw quad 2 half + 1 -> mid plane
        x0[sl] = *(const floatx4*)p0[sl];
        x1[sl] = *(const floatx4*)p1[sl];
      }
#pragma unroll
      for (int sl = 0; sl < 3; ++sl) {
        uintx2 h0, m0, h1, m1;
        img_split(x0[sl], h0, m0);
        img_split(x1[sl], h1, m1);
        if (sl < 2 || t4 + 512 < 2 * NPIX) {
          *(uintx4*)p0[sl] = uintx4{h0[0], h0[1], h1[0], h1[1]};
          *(uintx4*)p1[sl] = uintx4{m0[0], m0[1], m1[0], m1[1]};
        }
      }
    }
    if (splitter && phi < S && !(IMG_ABL & 2) && (!(IMG_ABL & 32) || phi == 0)) {
      // the B fragments of chunk phi, for this wave's contraction in the next phase
      const unsigned char* wts = wl + pc * WCH + ((q & 1) * NCOL + (r < NCOL ? r : r - NCOL)) * 16;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) Bf[tap][pl] = *(const bf16x8*)(wts + (tap * 2 + pl) * (2 * NCOL * 16));
    }
    if (!splitter && phi >= 1 && phi <= S && !(IMG_ABL & 2)) {
      // ---- contraction of chunk phi - 1: rows 4 wv .. 4 wv + 3 ----------------------------------------------
      const unsigned char* buf = ring + ((phi - 1) % NBUF) * CB;
#pragma unroll
      for (int rho = 0; rho < 6; ++rho)
#pragma unroll
        for (int wc = 0; wc < 3; ++wc) {
          const bf16x8 A = *(const bf16x8*)(buf + aoff[rho * 3 + wc]);
#pragma unroll
          for (int y = 0; y < 4; ++y) {
            const int wr = rho - y;
            if (wr >= 0 && wr < 3) {
              acc[y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, Bf[wr * 3 + wc][0], acc[y], 0, 0, 0);
              acc[y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, Bf[wr * 3 + wc][1], acc[y], 0, 0, 0);
            }
          }
        }
    }
    ISTAMP(2);
    if (splitter) { if (LEAD == 3 || fin0 || fin1) wait_older(newer); }
    else if (LEAD == 2) wait_older(0);
    ISTAMP(3);
    if (fin1) {
      if (colok && !(IMG_ABL & 8)) {
#pragma unroll
        for (int y = 0; y < 4; ++y) {
          floatx4 v = gather(acc[y]);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = __fadd_rn(prev[y][e], v[e]);
          const OutPos o = out_pos(pt - 1, y);
          if (o.gy < OH) {
            if (vec && o.gx + 3 < OW) *(floatx4*)o.p = v;
            else
              for (int e = 0; e < 4; ++e)
                if (o.gx + e < OW) o.p[e] = v[e];
          }
        }
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[m] = floatx4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    if (++pc == nch) { pc = 0; ++pt; }
    if (++dc == nch) { dc = 0; ++dt; }
  }
#if IMG_STAMP
  if (blockIdx.x == 0 && (tid == 0 || tid == 256))
    for (int i = 0; i < 64; ++i) ((unsigned*)a.out)[(tid >> 8) * 64 + i] = st[i];
#endif
}

// w [Cin][Cimg][5][5] (nn.ConvTranspose2d layout) -> dst floats [0, 9 Cin 16): [9 taps][Cin/16][16 columns][16 channels]
// fp32 (convT_image_kernel); the same number of bytes behind them: the two bf16 planes of the same values,
// [Cin/16][9 taps][plane: hi, mid][half][16 columns][8 channels] - the LDS image of convT_image_dma_kernel.
__device__ __forceinline__ float convT_image_g(const float* __restrict__ w, int Cin, int Cimg, int t, int ci, int col) {
  const int wr = t / 3, wc = t % 3;
  if (col < 4 * Cimg && ci < Cin) {
    const int c = col % Cimg, ph = col / Cimg;
    const int py = ph >> 1, px = ph & 1;
    if (wr >= py && wc >= px) return w[(((int64_t)ci * Cimg + c) * 5 + (py + 4 - 2 * wr)) * 5 + (px + 4 - 2 * wc)];
  }
  return 0.f;
}
__global__ void pack_convT_image_weight_kernel(const float* __restrict__ w, float* __restrict__ dst, int Cin,
                                               int Cimg, int C16, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  {
    const int ch = i & 15;
    int64_t rr = i >> 4;
    const int col = rr & 15;
    rr >>= 4;
    const int c16 = rr % C16;
    const int t = rr / C16;  // 0..8
    dst[i] = convT_image_g(w, Cin, Cimg, t, c16 * 16 + ch, col);
  }
  {
    const int e = i & 7, col = (i >> 3) & 15, half = (i >> 7) & 1;
    const int t = (int)((i >> 8) % 9), c16 = (int)((i >> 8) / 9);
    const float v = convT_image_g(w, Cin, Cimg, t, c16 * 16 + half * 8 + e, col);
    const __bf16 hi = (__bf16)v;                       // round to nearest even, as img_split
    const __bf16 mid = (__bf16)(v - (float)hi);
    __bf16* const sp = (__bf16*)(dst + total);
    const int64_t base = ((int64_t)(c16 * 9 + t) * 2) * 2 * 16 * 8 + (half * 16 + col) * 8 + e;
    sp[base] = hi;
    sp[base + 2 * 16 * 8] = mid;
  }
}

}  // namespace dsic

using namespace dsic;

extern "C" int64_t dsic_convT_image_weight_floats(int Cin) { return (int64_t)2 * 9 * Cin * 16; }

extern "C" int dsic_pack_convT_image_weight(const float* w, float* dst, int Cin, int Cimg, void* stream) {
  DSIC_REQUIRE(w && dst, "pack_convT_image_weight: null pointer");
  DSIC_REQUIRE(Cin > 0 && Cin % 16 == 0 && Cimg >= 1 && Cimg <= 4,
               "pack_convT_image_weight: Cin=%d must be a multiple of 16 and Cimg=%d in [1,4]", Cin, Cimg);
  const int64_t total = (int64_t)9 * Cin * 16;   // fp32 values; as many bytes again for the bf16 planes
  hipLaunchKernelGGL(pack_convT_image_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, w, dst, Cin, Cimg, Cin / 16, total);
  return check_launch("pack_convT_image_weight");
}

extern "C" int dsic_conv_transpose2d_image(const float* in, const float* w_packed, const float* bias,
                                           float* out_nchw, int B, int H, int W, int Cin, int Cimg,
                                           void* stream) {
  DSIC_REQUIRE(in && w_packed && bias && out_nchw, "convT_image: null pointer");
  DSIC_REQUIRE(B > 0 && H > 0 && W > 0, "convT_image: empty tensor");
  DSIC_REQUIRE(Cin > 0 && Cin % 16 == 0, "convT_image: Cin=%d must be a positive multiple of 16", Cin);
  DSIC_REQUIRE(Cimg >= 1 && Cimg <= 4, "convT_image: Cimg=%d not in [1,4]", Cimg);
  DSIC_REQUIRE((int64_t)H * W * Cin < (int64_t)1 << 29, "convT_image: image too large");
  ImgArgs a{};
  a.in = in; a.w = w_packed; a.bias = bias; a.out = out_nchw;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cimg = Cimg;
  a.tiles_x = ceil_div(W, IT_W); a.tiles_y = ceil_div(H, IT_H);
  const int64_t nblk = (int64_t)a.tiles_x * a.tiles_y * B;
  DSIC_REQUIRE(nblk < ((int64_t)1 << 31), "convT_image: grid too large");
  // DSIC_WINO_BF16=0 (the switch of the split-bf16 contractions, see conv_wino_bf16.hip) keeps this layer on
  // the fp32-input MFMA as well
  static int use_bf16 = -1;
  if (use_bf16 < 0) {
    const char* e = getenv("DSIC_WINO_BF16");
    use_bf16 = (e && e[0] == '0' && e[1] == 0) ? 0 : 1;
  }
  // Cin <= 128 (all weights fit in LDS beside the window ring) and an even number of 16-channel chunks (the two
  // wave groups take turns); DSIC_IMG_DMA=0 keeps the staged kernel, for A/B runs
  static int use_dma = -1;
  if (use_dma < 0) {
    const char* e = getenv("DSIC_IMG_DMA");
    use_dma = (e && e[0] == '0' && e[1] == 0) ? 0 : 1;
  }
  if (use_bf16 && use_dma && Cin <= 16 * imgd::MAXCH && Cin % 32 == 0) {
    a.tiles_x = ceil_div(W, imgd::TW); a.tiles_y = ceil_div(H, imgd::TH);
    const int64_t nt = (int64_t)a.tiles_x * a.tiles_y * B;
    DSIC_REQUIRE(nt < ((int64_t)1 << 30), "convT_image: too many tiles");
    a.ntiles = (int)nt;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    static int cus[64] = {};
    if (cus[dev] == 0) {
      int n = 0;
      if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
      const hipError_t e1 = hipFuncSetAttribute((const void*)convT_image_dma_kernel<5, 12>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                imgd::lds_bytes(16 * imgd::MAXCH, 5, 12));
      const hipError_t e2 = hipFuncSetAttribute((const void*)convT_image_dma_kernel<4, 16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                imgd::lds_bytes(16 * imgd::MAXCH, 4, 16));
      if (e1 != hipSuccess || e2 != hipSuccess) {
        set_error("convT_image: hipFuncSetAttribute: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
        return DSIC_EHIP;
      }
      cus[dev] = n;
    }
    const int grid = nt < cus[dev] ? (int)nt : cus[dev];
    if (Cimg <= 3)
      hipLaunchKernelGGL((convT_image_dma_kernel<5, 12>), dim3(grid), dim3(imgd::THREADS), imgd::lds_bytes(Cin, 5, 12), (hipStream_t)stream, a);
    else
      hipLaunchKernelGGL((convT_image_dma_kernel<4, 16>), dim3(grid), dim3(imgd::THREADS), imgd::lds_bytes(Cin, 4, 16), (hipStream_t)stream, a);
  } else if (use_bf16)
    hipLaunchKernelGGL(convT_image_kernel<true>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(convT_image_kernel<false>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("convT_image");
}
