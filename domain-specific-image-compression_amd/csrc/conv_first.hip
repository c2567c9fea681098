// First analysis layer: conv(Cimg, 128, 3, 1) + GDN (code/modelv2/layers.py:51)
// straight from the NCHW image, as an fp32 MFMA GEMM with K = 9*Cimg (27 -> 28,
// or 36 for 4 bands) instead of the generic kernel's 8-channel-padded K = 72.
//
// The layer is bound by its 512-byte-per-pixel NHWC output (33.5 MB per 256^2
// image), not by MACs: 14 MFMA k-steps per 32x32 tile, weights (28 x 128) held
// in registers for the whole kernel, the 3 x 10 x 18 input window of a 16x8
// pixel tile in LDS, A operands gathered with one ds_read_b32 per k-step
// (k = c*9 + ky*3 + kx, the reference weight order), GDN epilogue + LDS
// transpose + 16-byte stores as in conv_igemm.hip.
//
// BF16 = true (the default, DSIC_WINO_BF16 != 0): the contraction runs on v_mfma_f32_32x32x16_bf16 with both operands
// split into two bf16 planes (hi*hi + hi*mid + mid*hi, fp32 accumulate - the scheme of conv_wino_bf16.hip).  The
// window is expanded once per workgroup into an im2col tile in LDS, 128 pixels x [hi K' | mid K'] bf16 (K' = K
// rounded up to 16; pixel records 16 bytes longer than that: conflict-free ds_read_b128), so an A fragment is one
// 16-byte read per lane and plane, shared by the four waves; the weights are split once per wave into registers.
// 6 (K = 27) or 9 (K = 36) MFMAs of 32 cycles per 32x32 tile instead of 14 or 18 of 64.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace dsic {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// eight fp32 values -> bf16 planes hi = bf16(v), mid = bf16(v - hi) (round to nearest even)
__device__ __forceinline__ void first_split8(const floatx8 v, bf16x8& hi, bf16x8& mid) {
  hi = __builtin_convertvector(v, bf16x8);
  const floatx8 back = __builtin_convertvector(hi, floatx8);
  mid = __builtin_convertvector(v - back, bf16x8);
}

// U8: the image arrives as uint8 HWC (a decoded PNG/JPEG, PIL / numpy layout) and is converted on
// the fly exactly like torchvision's to_tensor (code/modelv2/modelseval.py:66-67): float(v) / 255.0f.
template <int C, bool U8 = false, bool BF16 = false>
__global__ __launch_bounds__(256, 2) void conv_first_kernel(
    const void* __restrict__ xv, const float* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ beta, const float* __restrict__ gamma, float* __restrict__ out, int B,
    int H, int W, int Cout, int act, int tiles_x, int tiles_y, int s2d) {
  constexpr int K = 9 * C;
  constexpr int KP = (K + 1) / 2 * 2;  // even
  constexpr int NS = KP / 2;           // MFMA k-steps
  constexpr int TW = 16, TH = 8, WW = TW + 2, WH = TH + 2, RS = 20;  // LDS row stride
  constexpr int EPI_STRIDE = 36;
  constexpr int WIN_FLOATS = C * WH * RS;
  constexpr int EPI_FLOATS = 4 * 32 * EPI_STRIDE;
  __shared__ __attribute__((aligned(16))) float lds[WIN_FLOATS > EPI_FLOATS ? WIN_FLOATS : EPI_FLOATS];
  constexpr int KB = (K + 15) / 16 * 16;   // bf16 path: K rounded to MFMA k-steps of 16
  constexpr int NKS = KB / 16;
  constexpr int PXB = 4 * KB + 16;         // bytes per im2col pixel record: hi[KB] | mid[KB] bf16 + pad
  __shared__ __attribute__((aligned(16))) unsigned char im2col[BF16 ? 128 * PXB : 16];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  int bx = blockIdx.x;
  const int tile_x = bx % tiles_x;
  bx /= tiles_x;
  const int tile_y = bx % tiles_y;
  const int n = bx / tiles_y;
  const int ox0 = tile_x * TW, oy0 = tile_y * TH;

  // weights of this wave's 32 output columns: b[s] = Wk[2s+h][col]
  const int col = wave * 32 + l31;
  const bool colok = col < Cout;
  float b[BF16 ? 1 : NS];
  int aoff[BF16 ? 1 : NS];   // per-lane LDS offsets of the A gather
  int abase[4];
  bf16x8 bhi[BF16 ? NKS : 1], bmid[BF16 ? NKS : 1];   // bf16 path: B fragments, lane (n = l31, k block h)
  if (BF16) {
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      floatx8 wv;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = ks * 16 + h * 8 + j;
        wv[j] = (colok && k < K) ? w[(size_t)col * K + k] : 0.f;
      }
      first_split8(wv, bhi[ks], bmid[ks]);
    }
  } else {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int k = 2 * s + h;
      b[s] = (colok && k < K) ? w[(size_t)col * K + k] : 0.f;  // reference layout [Cout][C][3][3]
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      int k = 2 * s + h;
      if (k >= K) k = 0;  // padded k-step: weight is zero, any valid address
      const int c = k / 9, t = k % 9;
      aoff[s] = (c * WH + t / 3) * RS + t % 3;
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int r = m * 32 + l31;
      abase[m] = (r / TW) * RS + r % TW;
    }
  }

  // stage the window (zero padded)
  if (U8) {
    const unsigned char* xin = (const unsigned char*)xv + (size_t)n * C * H * W;
    for (int i = tid; i < C * WH * WW; i += 256) {
      const int c = i % C, xx = (i / C) % WW, r = i / (C * WW);   // channel fastest: contiguous HWC bytes
      const int gy = oy0 - 1 + r, gx = ox0 - 1 + xx;
      float v = 0.f;
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = (float)xin[((size_t)gy * W + gx) * C + c] / 255.0f;
      lds[(c * WH + r) * RS + xx] = v;
    }
  } else {
    const float* xin = (const float*)xv + (size_t)n * C * H * W;
    for (int i = tid; i < C * WH * WW; i += 256) {
      const int c = i / (WH * WW), r = (i / WW) % WH, xx = i % WW;
      const int gy = oy0 - 1 + r, gx = ox0 - 1 + xx;
      float v = 0.f;
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = xin[((size_t)c * H + gy) * W + gx];
      lds[(c * WH + r) * RS + xx] = v;
    }
  }
  __syncthreads();

  floatx16 acc[4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;
  if (BF16) {
    // im2col: thread = (pixel, half of the k octets); octet o holds k = 8o .. 8o+7, k = c*9 + ky*3 + kx
    const int px = tid & 127, half = tid >> 7;   // half is wave-uniform
    const int pbase = (px / TW) * RS + px % TW;
    auto build = [&](auto half_tag) {   // the k -> (c, ky, kx) arithmetic folds at compile time
      constexpr int HALF = decltype(half_tag)::value;
#pragma unroll
      for (int oo = 0; oo < NKS; ++oo) {
        const int o = 2 * oo + HALF;
        floatx8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = 8 * o + j;
          const int c = k / 9, t = k % 9;
          v[j] = k < K ? lds[pbase + (c * WH + t / 3) * RS + t % 3] : 0.f;
        }
        bf16x8 vh, vm;
        first_split8(v, vh, vm);
        *(bf16x8*)(im2col + px * PXB + o * 16) = vh;
        *(bf16x8*)(im2col + px * PXB + 2 * KB + o * 16) = vm;
      }
    };
    if (half == 0)
      build(std::integral_constant<int, 0>{});
    else
      build(std::integral_constant<int, 1>{});
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const unsigned char* ap = im2col + (m * 32 + l31) * PXB + h * 16;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const bf16x8 ah = *(const bf16x8*)(ap + ks * 32);
        const bf16x8 am = *(const bf16x8*)(ap + 2 * KB + ks * 32);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bmid[ks], acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bhi[ks], acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bhi[ks], acc[m], 0, 0, 0);
      }
    }
  } else {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      float a[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) a[m] = lds[abase[m] + aoff[s]];
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[s], acc[m], 0, 0, 0);
    }
  }
  __syncthreads();  // window no longer needed: LDS becomes the transpose scratch

  const float bs = colok ? bias[col] : 0.f;
  float be = 1.f, ga = 0.f;
  if (colok && act == DSIC_ACT_GDN) {
    be = beta[col];
    ga = gamma[col];
  }
  float* epi = lds + wave * (32 * EPI_STRIDE);
#pragma unroll
  for (int m = 0; m < 4; ++m) {
#pragma unroll
    for (int e = 0; e < 16; e += 2) {  // pairs: packed fp32 arithmetic, this lane's column in both halves
      dsic_float2 v = {acc[m][e], acc[m][e + 1]};
      v = v + dsic_float2{bs, bs};
      if (act == DSIC_ACT_GDN) {
        v = gdn_pair<false>(v, dsic_float2{be, be}, dsic_float2{ga, ga});
      } else if (act == DSIC_ACT_RELU) {
        v[0] = v[0] > 0.f ? v[0] : 0.f;
        v[1] = v[1] > 0.f ? v[1] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int rr = ((e + k) & 3) + 8 * ((e + k) >> 2) + 4 * h;
        epi[rr * EPI_STRIDE + l31] = v[k];
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    const int c4 = (lane & 7) * 4;
    const int nn = wave * 32 + c4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rr = (lane >> 3) + 8 * i;
      const floatx4 v = *(const floatx4*)(epi + rr * EPI_STRIDE + c4);
      const int r = m * 32 + rr;
      const int ox = ox0 + r % TW, oy = oy0 + r / TW;
      if (oy < H && ox < W && nn < Cout) {
        // s2d: pixel (oy,ox) becomes channel block (oy&1)*2+(ox&1) of pixel (oy/2,ox/2)
        const size_t o = s2d ? (((size_t)n * (H >> 1) + (oy >> 1)) * (W >> 1) + (ox >> 1)) * (4 * Cout) +
                                   ((oy & 1) * 2 + (ox & 1)) * Cout + nn
                             : (((size_t)n * H + oy) * W + ox) * Cout + nn;
        __builtin_nontemporal_store(v, (floatx4*)(out + o));
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
  }
}

// to_tensor (modelseval.py:66-67, eval_selfcontained.py:58-59): uint8 HWC image -> float32 CHW in [0,1]
__global__ void u8hwc_to_f32nchw_kernel(const unsigned char* __restrict__ in, float* __restrict__ out, int C,
                                        int HW, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over [B][C][HW]
  if (i >= total) return;
  const int64_t p = i % HW;
  const int64_t bc = i / HW;
  const int c = (int)(bc % C);
  const int64_t b = bc / C;
  out[i] = (float)in[(b * HW + p) * C + c] / 255.0f;
}

}  // namespace dsic

using namespace dsic;

extern "C" int dsic_image_u8hwc_to_f32nchw(const unsigned char* x_u8_nhwc, float* out_nchw, int B, int C, int H,
                                           int W, void* stream) {
  DSIC_REQUIRE(x_u8_nhwc && out_nchw && B > 0 && C > 0 && H > 0 && W > 0, "image_u8hwc_to_f32nchw: bad argument");
  const int64_t total = (int64_t)B * C * H * W;
  hipLaunchKernelGGL(u8hwc_to_f32nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     x_u8_nhwc, out_nchw, C, H * W, total);
  return check_launch("image_u8hwc_to_f32nchw");
}

static int conv_first_launch(const void* x, bool u8, const float* w_oihw, const float* bias, const float* beta,
                             const float* gamma, float* out_nhwc, int B, int Cimg, int H, int W, int Cout, int act,
                             int s2d, void* stream) {
  DSIC_REQUIRE(x && w_oihw && bias && out_nhwc, "conv_first: null pointer");
  DSIC_REQUIRE(B > 0 && H > 0 && W > 0, "conv_first: empty tensor");
  DSIC_REQUIRE(Cimg == 3 || Cimg == 4, "conv_first: Cimg=%d must be 3 or 4", Cimg);
  DSIC_REQUIRE(Cout > 0 && Cout <= 128 && Cout % 4 == 0, "conv_first: Cout=%d must be a multiple of 4, <= 128", Cout);
  DSIC_REQUIRE(act == DSIC_ACT_NONE || act == DSIC_ACT_GDN || act == DSIC_ACT_RELU, "conv_first: act=%d", act);
  DSIC_REQUIRE(act != DSIC_ACT_GDN || (beta && gamma), "conv_first: GDN needs beta and gamma");
  DSIC_REQUIRE(!s2d || (H % 2 == 0 && W % 2 == 0), "conv_first: space-to-depth output needs even H and W");
  const int tx = ceil_div(W, 16), ty = ceil_div(H, 8);
  DSIC_REQUIRE((int64_t)tx * ty * B < ((int64_t)1 << 31), "conv_first: grid too large");
  dim3 grid(tx * ty * B), block(256);
  hipStream_t st = (hipStream_t)stream;
  // dsic_set_split_bf16(0) / DSIC_WINO_BF16=0 (the switch that selects the exact-fp32 Winograd kernels) keeps this
  // layer on the fp32-input MFMA
  const int use_bf16 = split_bf16();
#define DSIC_FIRST(CC, UU, BB)                                                                                        \
  hipLaunchKernelGGL((conv_first_kernel<CC, UU, BB>), grid, block, 0, st, x, w_oihw, bias, beta, gamma, out_nhwc, B, H, \
                     W, Cout, act, tx, ty, s2d)
#define DSIC_FIRST2(CC, UU) \
  do {                      \
    if (use_bf16)           \
      DSIC_FIRST(CC, UU, true); \
    else                    \
      DSIC_FIRST(CC, UU, false); \
  } while (0)
  if (Cimg == 3) {
    if (u8) DSIC_FIRST2(3, true); else DSIC_FIRST2(3, false);
  } else {
    if (u8) DSIC_FIRST2(4, true); else DSIC_FIRST2(4, false);
  }
#undef DSIC_FIRST2
#undef DSIC_FIRST
  return check_launch("conv_first");
}

// conv(Cimg,Cout,3,1) (+GDN/ReLU) from the NCHW image to NHWC features
// (layers.py:51).  w is the REFERENCE weight tensor [Cout][Cimg][3][3], unpacked.
extern "C" int dsic_conv_first_nchw(const float* x_nchw, const float* w_oihw, const float* bias,
                                    const float* beta, const float* gamma, float* out_nhwc, int B,
                                    int Cimg, int H, int W, int Cout, int act, int s2d, void* stream) {
  return conv_first_launch(x_nchw, false, w_oihw, bias, beta, gamma, out_nhwc, B, Cimg, H, W, Cout, act, s2d, stream);
}

// The same layer straight from the decoded uint8 HWC image: to_tensor (modelseval.py:66-67) fused
// into the window staging, a quarter of the float image's bytes over PCIe and from HBM.
extern "C" int dsic_conv_first_u8hwc(const unsigned char* x_u8_nhwc, const float* w_oihw, const float* bias,
                                     const float* beta, const float* gamma, float* out_nhwc, int B, int Cimg,
                                     int H, int W, int Cout, int act, int s2d, void* stream) {
  return conv_first_launch(x_u8_nhwc, true, w_oihw, bias, beta, gamma, out_nhwc, B, Cimg, H, W, Cout, act, s2d, stream);
}
