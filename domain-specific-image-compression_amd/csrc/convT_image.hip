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
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

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

#ifndef IMG_TH
#define IMG_TH 16
#endif
constexpr int IT_W = 32, IT_H = IMG_TH;      // input-grid pixels per workgroup tile
constexpr int RPW = IT_H / 4, NM = 2 * RPW;  // tile rows and M tiles (16 pixels) per wave
constexpr int IW = IT_W + 2, IH = IT_H + 2;  // staged window
constexpr int ICK = 16, IP = ICK + 4;        // channels per chunk, LDS floats per window pixel
constexpr int ISLOTS = (IH * IW * (ICK / 4) + 255) / 256;  // float4 staging slots per thread
constexpr int IMG_WIN_BYTES = IH * IW * IP * 4;            // 48 960
constexpr int IMG_WCH = 9 * 1024;                          // bf16 weight planes of a 16-channel chunk

struct ImgArgs {
  const float* in;
  const float* w;  // packed [9][Cin/16][16][16]
  const float* bias;
  float* out;  // [B][Cimg][2H][2W]
  int B, H, W, Cin, Cimg;
  int tiles_x, tiles_y;
};

// timing-only ablations (wrong results; tools/run_img_abl.sh): 1 no global window loads, 2 no MFMAs, 4 no weight
// loads / splits, 8 no output stores, 16 no window split + LDS stores, 32 no LDS A reads
#ifndef IMG_ABL
#define IMG_ABL 0
#endif
#ifndef IMG_WGS
#define IMG_WGS 2  // workgroups per CU the register budget is set for (3 spills and is slower)
#endif
// BF16 = true: both operands are split into two bf16 planes (hi = bf16(v), mid = bf16(v - hi)) and the contraction
// runs on v_mfma_f32_16x16x32_bf16 with K = [hi 16 channels | mid 16 channels] of ONE 16-channel chunk: the LDS
// pixel record is 16 bf16 hi | 16 bf16 mid | pad (the same 80 bytes as 16 fp32 + pad; the window is split once
// while it is staged), so lane (pixel r, k-block q) reads its A operand as the 16 bytes at 16 q of the record
// (5 r + q mod 16: conflict-free), and with B = [Uhi | Uhi], then [Umid | Umid] (planes written by the pack kernel
// behind the fp32 values, two 16-byte loads per tap, no conversion in the loop) two MFMAs of 16 cycles give
// (Vhi + Vmid)(Uhi + Umid) - all four products of the two-plane split.  Round 2 used three
// v_mfma_f32_16x16x16_bf16 (hi*hi, hi*mid, mid*hi; 48 cycles), two 8-byte LDS reads per fragment and split
// the fp32 weight quads in the loop: 0.285 -> 0.2xx ms.
template <bool BF16>
__global__ __launch_bounds__(256, IMG_WGS) void convT_image_kernel(const ImgArgs a) {
  // window / output tile (48 960 B); BF16: + two 9 KB buffers with the bf16 weight planes of a chunk
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  float* const lds = (float*)lds_raw;
  const unsigned char* const wlds = lds_raw + IMG_WIN_BYTES;
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

  floatx4 acc[NM];
#pragma unroll
  for (int m = 0; m < NM; ++m) acc[m] = floatx4{0.f, 0.f, 0.f, 0.f};
  // A fragment of M tile m (row m>>1 of this wave, x half m&1) at tap (0,0): + (wr*IW + wc)*IP per tap
  const int abase = ((wave * RPW) * IW + r) * IP + 4 * q;
  const float* wl = a.w + r * 16 + 4 * q;  // + (tap*C16 + chunk)*256
  // bf16 planes of the weights: [chunk][tap][plane][half][col 16][8 channels] behind the 9 Cin 16 fp32 values.  A
  // chunk's 9 KB go global -> LDS by buffer_load_dwordx4 ... lds (wave w: the 1 KB blocks of taps w, w + 4, w + 8;
  // no registers), one chunk ahead into the other of two buffers: the MFMA loop then holds no global load at all.
  // (Fetched from L1 / L2 inside the loop, one tap ahead, every weight load queued behind the window loads of
  // load_pair in the in-order vector-memory path: -0.05 ms for the layer.)  The loads are inline asm, invisible
  // to the compiler; they are waited for by vmcnt(0) in front of the second barrier of their chunk, where the only
  // other loads in flight are window loads issued before them.
  const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)lds_raw;
  const unsigned long long wsb = (unsigned long long)(a.w + (size_t)9 * Cin * 16);
  const uintx4 wrs = {(unsigned)wsb, (unsigned)(wsb >> 32) & 0xFFFFu, (unsigned)(9 * Cin * 16 * 4), 0x00020000u};
  auto weights_dma = [&](int chunk) {
    if (!BF16 || (IMG_ABL & 4)) return;
    const unsigned voff = (unsigned)lane * 16u;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int tap = wave + 4 * k;
      if (tap < 9) {
        const unsigned m0v = lds_base + (unsigned)(IMG_WIN_BYTES + (chunk & 1) * IMG_WCH + tap * 1024);
        const unsigned soff = (unsigned)((chunk * 9 + tap) * 1024);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                     :: "s"(m0v), "v"(voff), "s"(wrs), "s"(soff) : "memory");
      }
    }
  };

  weights_dma(0);
  load_pair(0);
  for (int chunk = 0; chunk < C16; ++chunk) {
    __syncthreads();  // the previous chunk's window is no longer read
    store_chunk(chunk & 1);
    if (BF16) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this chunk's weights (and the window pair) have landed
    __syncthreads();
    if ((chunk & 1) && chunk + 1 < C16) load_pair(chunk + 1);  // in flight during the MFMAs below
    if (chunk + 1 < C16) weights_dma(chunk + 1);               // the other buffer was read last in chunk - 1
    if (BF16) {
      // lane (col r, q) takes half q & 1 of plane hi (B1) and of plane mid (B2)
      const unsigned char* wc0 = wlds + (chunk & 1) * IMG_WCH + ((q & 1) * 16 + r) * 16;
      // A fragment of lane (r, q): bytes 16 q .. of the record of pixel r of the M tile
      const unsigned char* abyte = (const unsigned char*)lds + (size_t)((wave * RPW) * IW + r) * (IP * 4) + q * 16;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int wr = tap / 3, wc = tap % 3;
        bf16x8 B1, B2;
        if (IMG_ABL & 4) { B1 = bf16x8{1, 2, 3, 4, 5, 6, 7, 8}; B2 = B1; }
        else { B1 = *(const bf16x8*)(wc0 + tap * 1024); B2 = *(const bf16x8*)(wc0 + tap * 1024 + 512); }
#pragma unroll
        for (int m = 0; m < NM; ++m) {
          const unsigned char* ap = abyte + (size_t)((((m >> 1) + wr) * IW + (m & 1) * 16 + wc) * (IP * 4));
          bf16x8 A;
          if (IMG_ABL & 32) A = B1;
          else A = *(const bf16x8*)ap;
          if (IMG_ABL & 2) {
            acc[m][0] += (float)A[0] + (float)B1[1] + (float)B2[2];
          } else {
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B1, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B2, acc[m], 0, 0, 0);
          }
        }
      }
    } else {
      floatx4 b = *(const floatx4*)(wl + (size_t)(0 * C16 + chunk) * 256);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int wr = tap / 3, wc = tap % 3;
        const floatx4 bc = b;
        if (tap + 1 < 9) b = *(const floatx4*)(wl + (size_t)((tap + 1) * C16 + chunk) * 256);
#pragma unroll
        for (int m = 0; m < NM; ++m) {
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
    for (int m = 0; m < NM; ++m) {
      const int yl = wave * RPW + (m >> 1);
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

// w [Cin][Cimg][5][5] (nn.ConvTranspose2d layout) -> dst floats [0, 9 Cin 16): [9 taps][Cin/16][16 columns][16 channels]
// fp32 (the fp32-input MFMA path); the same number of bytes behind them: the two bf16 planes of the same values,
// [Cin/16][9 taps][plane: hi, mid][half][16 columns][8 channels] (the B operands of the split-bf16 path).
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
  // dsic_set_split_bf16(0) / DSIC_WINO_BF16=0 (the switch of the split-bf16 contractions, see conv_wino_bf16.hip)
  // keeps this layer on the fp32-input MFMA as well
  const int use_bf16 = split_bf16();
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  static bool attr_set[64] = {};
  if (!attr_set[dev]) {   // 48 960 + 18 432 bytes of dynamic LDS exceed the 64 KB a kernel gets without asking
    const hipError_t e = hipFuncSetAttribute((const void*)convT_image_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             IMG_WIN_BYTES + 2 * IMG_WCH);
    if (e != hipSuccess) {
      set_error("convT_image: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return DSIC_EHIP;
    }
    attr_set[dev] = true;
  }
  if (use_bf16)
    hipLaunchKernelGGL(convT_image_kernel<true>, dim3((unsigned)nblk), dim3(256), IMG_WIN_BYTES + 2 * IMG_WCH, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(convT_image_kernel<false>, dim3((unsigned)nblk), dim3(256), IMG_WIN_BYTES, (hipStream_t)stream, a);
  return check_launch("convT_image");
}
