// Weight re-layout and NCHW<->NHWC helpers (run once per checkpoint load or at
// the NCHW boundary of the reference API; none of them is on the timed path
// except image_to_nhwc8 and the small latent transposes).
#include "common.h"

namespace dsic {

__global__ void pack_conv_weight_kernel(const float* __restrict__ w, float* __restrict__ dst,
                                        int Cout, int Cin, int k, int Cin8, int CoutP,
                                        int64_t total) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int j = i & 7;
  int64_t r = i >> 3;
  const int n = r % CoutP;
  r /= CoutP;
  const int c8 = r % Cin8;
  const int t = r / Cin8;
  const int c = c8 * 8 + j;
  const int ky = t / k, kx = t % k;
  float v = 0.f;
  if (n < Cout && c < Cin) v = w[(((int64_t)n * Cin + c) * k + ky) * k + kx];
  dst[i] = v;
}

// phase p = py*2+px owns taps [base[p], base[p]+(3-py)(3-px)); local tap
// (ty,tx) reads input (oy-1+ty+py, ox-1+tx+px) and kernel element
// ky = py+4-2(ty+py), kx = px+4-2(tx+px)  (from oy = 2*iy - 2 + ky).
__global__ void pack_convT_weight_kernel(const float* __restrict__ w, float* __restrict__ dst,
                                         int Cin, int Cout, int Cin8, int CoutP, int64_t total) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int j = i & 7;
  int64_t r = i >> 3;
  const int n = r % CoutP;
  r /= CoutP;
  const int c8 = r % Cin8;
  const int T = r / Cin8;  // 0..24
  const int phase = T < 9 ? 0 : (T < 15 ? 1 : (T < 21 ? 2 : 3));
  const int base = phase == 0 ? 0 : (phase == 1 ? 9 : (phase == 2 ? 15 : 21));
  const int py = phase >> 1, px = phase & 1;
  const int ntx = 3 - px;
  const int tl = T - base;
  const int ty = tl / ntx, tx = tl % ntx;
  const int ky = py + 4 - 2 * (ty + py), kx = px + 4 - 2 * (tx + px);
  const int c = c8 * 8 + j;
  float v = 0.f;
  if (n < Cout && c < Cin) v = w[(((int64_t)c * Cout + n) * 5 + ky) * 5 + kx];
  dst[i] = v;
}

__global__ void image_to_nhwc8_kernel(const float* __restrict__ x, float* __restrict__ dst, int C,
                                      int HW, int64_t total) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over B*HW pixels
  if (i >= total) return;
  const int64_t b = i / HW;
  const int p = i % HW;
  float v[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) v[c] = c < C ? x[(b * C + c) * HW + p] : 0.f;
  float4* o = (float4*)(dst + i * 8);
  o[0] = make_float4(v[0], v[1], v[2], v[3]);
  o[1] = make_float4(v[4], v[5], v[6], v[7]);
}

// 32x32 tile transpose through LDS: src viewed as [R][Cc] per batch -> [Cc][R].
__global__ void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int R,
                                 int Cc) {
  __shared__ float tile[32][33];
  const int64_t b = blockIdx.z;
  const float* s = src + b * (int64_t)R * Cc;
  float* d = dst + b * (int64_t)R * Cc;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + threadIdx.x;
    if (r < R && c < Cc) tile[i][threadIdx.x] = s[(int64_t)r * Cc + c];
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + threadIdx.x;
    if (r < R && c < Cc) d[(int64_t)c * R + r] = tile[threadIdx.x][i];
  }
}

// F.pad(x, (0,pad_w,0,pad_h), mode="reflect") of modelseval.py:57-64 (bottom/right only).
__global__ void reflect_pad_br_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W,
                                      int Hp, int Wp, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int x = i % Wp;
  const int y = (i / Wp) % Hp;
  const int64_t plane = i / ((int64_t)Wp * Hp);
  const int sy = y < H ? y : 2 * (H - 1) - y;  // reflect without repeating the edge
  const int sx = x < W ? x : 2 * (W - 1) - x;
  dst[i] = src[(plane * H + sy) * W + sx];
}

// combinebandsall.py:7-12,35-36: per band  b -= min; b /= max (if max != 0); rgb = uint8(b*255).
// One workgroup per (image, band) plane; out01 keeps the normalised float plane (the model
// input), out_u8 (optional) the truncated 8-bit value the reference writes to PNG.
__global__ __launch_bounds__(256) void normalize_bands_kernel(const float* __restrict__ src,
                                                              float* __restrict__ out01,
                                                              uint8_t* __restrict__ out_u8, int HW) {
  __shared__ float red[2][4];
  const float* p = src + (size_t)blockIdx.x * HW;
  float mn = p[0], mx = p[0];
  for (int i = threadIdx.x; i < HW; i += 256) {
    const float v = p[i];
    mn = fminf(mn, v);
    mx = fmaxf(mx, v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mn = fminf(mn, __shfl_down(mn, o, 64));
    mx = fmaxf(mx, __shfl_down(mx, o, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = mn;
    red[1][threadIdx.x >> 6] = mx;
  }
  __syncthreads();
  mn = fminf(fminf(red[0][0], red[0][1]), fminf(red[0][2], red[0][3]));
  mx = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
  const float range = mx - mn;  // band.max() after the subtraction
  for (int i = threadIdx.x; i < HW; i += 256) {
    float v = p[i] - mn;
    if (range != 0.f) v = v / range;
    out01[(size_t)blockIdx.x * HW + i] = v;
    if (out_u8) out_u8[(size_t)blockIdx.x * HW + i] = (uint8_t)(v * 255.f);
  }
}

static int transpose_launch(const float* src, float* dst, int B, int R, int Cc, hipStream_t st) {
  dim3 grid(ceil_div(Cc, 32), ceil_div(R, 32), B), block(32, 8);
  hipLaunchKernelGGL(transpose_kernel, grid, block, 0, st, src, dst, R, Cc);
  return check_launch("transpose");
}

}  // namespace dsic

using namespace dsic;

extern "C" int64_t dsic_packed_conv_weight_floats(int Cout, int Cin, int k) {
  return (int64_t)k * k * (round_up(Cin, 8) / 8) * round_up(Cout, 32) * 8;
}

extern "C" int dsic_pack_conv_weight(const float* w, float* dst, int Cout, int Cin, int k,
                                     void* stream) {
  DSIC_REQUIRE(w && dst, "pack_conv_weight: null pointer");
  DSIC_REQUIRE(Cout > 0 && Cin > 0 && (k == 1 || k == 3 || k == 5), "pack_conv_weight: bad shape");
  const int Cin8 = round_up(Cin, 8) / 8, CoutP = round_up(Cout, 32);
  const int64_t total = dsic_packed_conv_weight_floats(Cout, Cin, k);
  hipLaunchKernelGGL(pack_conv_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, w, dst, Cout, Cin, k, Cin8, CoutP, total);
  return check_launch("pack_conv_weight");
}

extern "C" int dsic_pack_convT_weight(const float* w, float* dst, int Cin, int Cout,
                                      void* stream) {
  DSIC_REQUIRE(w && dst, "pack_convT_weight: null pointer");
  DSIC_REQUIRE(Cout > 0 && Cin > 0 && Cin % 8 == 0, "pack_convT_weight: Cin must be a multiple of 8");
  const int Cin8 = Cin / 8, CoutP = round_up(Cout, 32);
  const int64_t total = (int64_t)25 * Cin8 * CoutP * 8;
  hipLaunchKernelGGL(pack_convT_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, w, dst, Cin, Cout, Cin8, CoutP, total);
  return check_launch("pack_convT_weight");
}

extern "C" int dsic_image_to_nhwc8(const float* x, float* dst, int B, int C, int H, int W,
                                   void* stream) {
  DSIC_REQUIRE(x && dst, "image_to_nhwc8: null pointer");
  DSIC_REQUIRE(B > 0 && C >= 1 && C <= 8 && H > 0 && W > 0, "image_to_nhwc8: bad shape");
  const int64_t total = (int64_t)B * H * W;
  hipLaunchKernelGGL(image_to_nhwc8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, x, dst, C, H * W, total);
  return check_launch("image_to_nhwc8");
}

extern "C" int dsic_nhwc_to_nchw(const float* src, float* dst, int B, int H, int W, int C,
                                 void* stream) {
  DSIC_REQUIRE(src && dst && B > 0 && H > 0 && W > 0 && C > 0, "nhwc_to_nchw: bad argument");
  return transpose_launch(src, dst, B, H * W, C, (hipStream_t)stream);
}

extern "C" int dsic_nchw_to_nhwc(const float* src, float* dst, int B, int C, int H, int W,
                                 void* stream) {
  DSIC_REQUIRE(src && dst && B > 0 && H > 0 && W > 0 && C > 0, "nchw_to_nhwc: bad argument");
  return transpose_launch(src, dst, B, C, H * W, (hipStream_t)stream);
}

extern "C" int dsic_reflect_pad_br(const float* src, float* dst, int planes, int H, int W, int pad_h,
                                   int pad_w, void* stream) {
  DSIC_REQUIRE(src && dst && planes > 0 && H > 0 && W > 0 && pad_h >= 0 && pad_w >= 0, "reflect_pad: bad argument");
  DSIC_REQUIRE(pad_h < H && pad_w < W, "reflect_pad: padding (%d,%d) must be smaller than the image (%d,%d)", pad_h, pad_w, H, W);
  const int Hp = H + pad_h, Wp = W + pad_w;
  const int64_t total = (int64_t)planes * Hp * Wp;
  hipLaunchKernelGGL(reflect_pad_br_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, src, dst, H, W, Hp, Wp, total);
  return check_launch("reflect_pad");
}

extern "C" int dsic_normalize_bands(const float* bands, float* out01, uint8_t* out_u8, int planes, int HW,
                                    void* stream) {
  DSIC_REQUIRE(bands && out01 && planes > 0 && HW > 0, "normalize_bands: bad argument");
  hipLaunchKernelGGL(normalize_bands_kernel, dim3(planes), dim3(256), 0, (hipStream_t)stream, bands, out01,
                     out_u8, HW);
  return check_launch("normalize_bands");
}
