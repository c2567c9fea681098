// Distortion metrics on the GPU: MS-SSIM / SSIM and MSE.
//
// MS-SSIM restates pytorch-msssim 1.0.0 (called at modelseval.py:78-88 and
// eval_selfcontained_entropy.py:154; third-party, not vendored in the
// reference, so parity is pinned only by the repo's own oracle):
// 11-tap Gaussian window (sigma 1.5), separable "valid" filtering of
// X, Y, X*X, Y*Y, X*Y (rows first, then columns), C1=(0.01 L)^2, C2=(0.03 L)^2,
// cs = (2 s_xy + C2)/(s_x + s_y + C2), ssim = (2 m_x m_y + C1)/(m_x^2+m_y^2+C1)*cs,
// spatial means per (image, channel), 2x2 average pooling (padding = size%2,
// zeros counted) between levels, prod_l relu(v_l)^w_l, mean over channels.
// One 32x16 output tile per workgroup, LDS-staged 42x26 halo tile, fixed-order fp64 partial
// sums (deterministic).
#include "common.h"

namespace dsic {

#define WIN 11
#define TSX 32                 // output tile: 32 x 16 per workgroup
#define TSY 16
#define TINX (TSX + WIN - 1)   // 42 x 26 input tile
#define TINY (TSY + WIN - 1)

struct GaussWin {
  float g[WIN];
};

// The scan is LDS-bound, not HBM-bound: every thread produces FOUR adjacent outputs along the
// filter direction from 14 values held in registers (sliding window), 4x fewer LDS reads than one
// output per thread.  Each output still sums its 11 taps in ascending order.
__global__ __launch_bounds__(256) void ssim_level_kernel(const float* __restrict__ X,
                                                         const float* __restrict__ Y,
                                                         double* __restrict__ partial, int H, int W,
                                                         int tiles_x, int tiles_y, float C1, float C2,
                                                         int clamp_x, GaussWin gw) {
  __shared__ float sx[TINY][TINX + 1], sy[TINY][TINX + 1];
  __shared__ float tmp[5][TSY][TINX + 1];
  __shared__ double red[2][4];
  const int plane = blockIdx.z;
  const int tx0 = blockIdx.x * TSX, ty0 = blockIdx.y * TSY;
  const int Ho = H - (WIN - 1), Wo = W - (WIN - 1);
  const float* xp = X + (size_t)plane * H * W;
  const float* yp = Y + (size_t)plane * H * W;
  const int tid = threadIdx.x;
  for (int i = tid; i < TINY * TINX; i += 256) {
    const int r = i / TINX, c = i % TINX;
    const int gy = ty0 + r, gx = tx0 + c;
    float a = 0.f, b = 0.f;
    if (gy < H && gx < W) {
      a = xp[(size_t)gy * W + gx];
      b = yp[(size_t)gy * W + gx];
      if (clamp_x) a = fminf(fmaxf(a, 0.f), 1.f);
    }
    sx[r][c] = a;
    sy[r][c] = b;
  }
  __syncthreads();
  // pass 1: filter along H -> tmp[map][out_row][col]; thread = (column, group of 4 output rows)
  if (tid < TINX * (TSY / 4)) {
    const int c = tid % TINX, r0 = (tid / TINX) * 4;
    float a[14], b[14], aa[14], bb[14], ab[14];
#pragma unroll
    for (int i = 0; i < 14; ++i) {
      a[i] = sx[r0 + i][c];
      b[i] = sy[r0 + i][c];
      aa[i] = a[i] * a[i];
      bb[i] = b[i] * b[i];
      ab[i] = a[i] * b[i];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f, m4 = 0.f;
#pragma unroll
      for (int k = 0; k < WIN; ++k) {
        const float g = gw.g[k];
        m0 += g * a[j + k];
        m1 += g * b[j + k];
        m2 += g * aa[j + k];
        m3 += g * bb[j + k];
        m4 += g * ab[j + k];
      }
      tmp[0][r0 + j][c] = m0;
      tmp[1][r0 + j][c] = m1;
      tmp[2][r0 + j][c] = m2;
      tmp[3][r0 + j][c] = m3;
      tmp[4][r0 + j][c] = m4;
    }
  }
  __syncthreads();
  // pass 2: filter along W; thread = (output row, group of 4 output columns)
  double cs = 0.0, ss = 0.0;
  if (tid < TSY * (TSX / 4)) {
    const int r = tid / (TSX / 4), c0 = (tid % (TSX / 4)) * 4;
    float m[5][4];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      float t[14];
#pragma unroll
      for (int i = 0; i < 14; ++i) t[i] = tmp[q][r][c0 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < WIN; ++k) s += gw.g[k] * t[j + k];
        m[q][j] = s;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (ty0 + r < Ho && tx0 + c0 + j < Wo) {
        const float m0 = m[0][j], m1 = m[1][j];
        const float mu1_sq = m0 * m0, mu2_sq = m1 * m1, mu12 = m0 * m1;
        const float s1 = m[2][j] - mu1_sq, s2 = m[3][j] - mu2_sq, s12 = m[4][j] - mu12;
        const float csv = (2.f * s12 + C2) / (s1 + s2 + C2);
        const float sv = ((2.f * mu12 + C1) / (mu1_sq + mu2_sq + C1)) * csv;
        cs += (double)csv;
        ss += (double)sv;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    cs += __shfl_down(cs, o, 64);
    ss += __shfl_down(ss, o, 64);
  }
  if ((tid & 63) == 0) {
    red[0][tid >> 6] = cs;
    red[1][tid >> 6] = ss;
  }
  __syncthreads();
  if (tid == 0) {
    const size_t o = ((size_t)plane * tiles_y * tiles_x + (size_t)blockIdx.y * tiles_x + blockIdx.x) * 2;
    partial[o] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    partial[o + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}

// means[plane][2] = (mean cs, mean ssim) from the tile partials, fixed order.
__global__ void ssim_reduce_kernel(const double* __restrict__ partial, double* __restrict__ means,
                                   int ntiles, double inv_count, int planes) {
  const int plane = blockIdx.x * blockDim.x + threadIdx.x;
  if (plane >= planes) return;
  double cs = 0.0, ss = 0.0;
  for (int t = 0; t < ntiles; ++t) {
    cs += partial[((size_t)plane * ntiles + t) * 2];
    ss += partial[((size_t)plane * ntiles + t) * 2 + 1];
  }
  means[2 * plane] = cs * inv_count;
  means[2 * plane + 1] = ss * inv_count;
}

__global__ void avgpool2_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W,
                                int Ho, int Wo, int ph, int pw, int clamp, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int ox = i % Wo;
  const int oy = (i / Wo) % Ho;
  const int64_t plane = i / ((int64_t)Wo * Ho);
  const float* s = src + plane * H * W;
  float acc = 0.f;
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const int y = 2 * oy - ph + dy, x = 2 * ox - pw + dx;
      if (y >= 0 && y < H && x >= 0 && x < W) {
        float v = s[(size_t)y * W + x];
        if (clamp) v = fminf(fmaxf(v, 0.f), 1.f);
        acc += v;
      }
    }
  dst[i] = acc * 0.25f;  // count_include_pad=True
}

// means: [levels][B*C][2]; out[b] = mean_c prod_l relu(v_l)^w_l with v_l = cs for
// l < levels-1 and ssim for the last level.  relu_last=0 gives plain SSIM.
__global__ void msssim_finalize_kernel(const double* __restrict__ means, const float* __restrict__ w,
                                       float* __restrict__ out, int levels, int B, int C,
                                       int relu_last) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float acc = 0.f;
  for (int c = 0; c < C; ++c) {
    float prod = 1.f;
    for (int l = 0; l < levels; ++l) {
      const double* m = means + ((size_t)l * B * C + (size_t)b * C + c) * 2;
      float v = (float)(l == levels - 1 ? m[1] : m[0]);
      if (l < levels - 1 || relu_last) v = fmaxf(v, 0.f);
      prod *= (levels == 1 && !relu_last) ? v : powf(v, w[l]);
    }
    acc += prod;
  }
  out[b] = acc / (float)C;
}

__global__ __launch_bounds__(256) void sqerr_kernel(const float* __restrict__ a,
                                                    const float* __restrict__ b,
                                                    double* __restrict__ out, int64_t n_per_image,
                                                    int clamp_a) {
  __shared__ double red[4];
  const int img = blockIdx.x;
  const float* pa = a + (size_t)img * n_per_image;
  const float* pb = b + (size_t)img * n_per_image;
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < n_per_image; i += 256) {
    float x = pa[i];
    if (clamp_a) x = fminf(fmaxf(x, 0.f), 1.f);
    const float d = x - pb[i];
    acc += (double)(d * d);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[img] = red[0] + red[1] + red[2] + red[3];
}

}  // namespace dsic

using namespace dsic;

extern "C" int dsic_ssim_level(const float* X, const float* Y, double* partial, double* means,
                               int planes, int H, int W, float C1, float C2, int clamp_x,
                               void* stream) {
  DSIC_REQUIRE(X && Y && partial && means, "ssim_level: null pointer");
  DSIC_REQUIRE(planes > 0 && H >= WIN && W >= WIN, "ssim_level: plane %dx%d smaller than the 11x11 window", H, W);
  const int Ho = H - (WIN - 1), Wo = W - (WIN - 1);
  const int tx = ceil_div(Wo, TSX), ty = ceil_div(Ho, TSY);
  DSIC_REQUIRE(planes <= 65535, "ssim_level: too many planes (%d)", planes);
  GaussWin gw;
  {
    // pytorch-msssim _fspecial_gauss_1d: exp(-(i-5)^2/(2*1.5^2)) normalised, in fp32
    float s = 0.f;
    for (int i = 0; i < WIN; ++i) {
      const float d = (float)(i - WIN / 2);
      gw.g[i] = expf(-(d * d) / (2.f * 1.5f * 1.5f));
      s += gw.g[i];
    }
    for (int i = 0; i < WIN; ++i) gw.g[i] /= s;
  }
  hipLaunchKernelGGL(ssim_level_kernel, dim3(tx, ty, planes), dim3(256), 0, (hipStream_t)stream, X, Y,
                     partial, H, W, tx, ty, C1, C2, clamp_x, gw);
  int rc = check_launch("ssim_level");
  if (rc) return rc;
  hipLaunchKernelGGL(ssim_reduce_kernel, dim3(ceil_div(planes, 64)), dim3(64), 0, (hipStream_t)stream,
                     partial, means, tx * ty, 1.0 / ((double)Ho * Wo), planes);
  return check_launch("ssim_reduce");
}

extern "C" int64_t dsic_ssim_partial_doubles(int planes, int H, int W) {
  if (H < WIN || W < WIN) return 0;
  return (int64_t)planes * ceil_div(H - WIN + 1, TSY) * ceil_div(W - WIN + 1, TSX) * 2;
}

extern "C" int dsic_avgpool2(const float* src, float* dst, int planes, int H, int W, int clamp,
                             void* stream) {
  DSIC_REQUIRE(src && dst && planes > 0 && H > 0 && W > 0, "avgpool2: bad argument");
  const int ph = H % 2, pw = W % 2;
  const int Ho = (H + 2 * ph - 2) / 2 + 1, Wo = (W + 2 * pw - 2) / 2 + 1;
  const int64_t total = (int64_t)planes * Ho * Wo;
  hipLaunchKernelGGL(avgpool2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, src, dst, H, W, Ho, Wo, ph, pw, clamp, total);
  return check_launch("avgpool2");
}

extern "C" int dsic_msssim_finalize(const double* means, const float* weights, float* out, int levels,
                                    int B, int C, int relu_last, void* stream) {
  DSIC_REQUIRE(means && weights && out && levels >= 1 && levels <= 8 && B > 0 && C > 0,
               "msssim_finalize: bad argument");
  hipLaunchKernelGGL(msssim_finalize_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, (hipStream_t)stream,
                     means, weights, out, levels, B, C, relu_last);
  return check_launch("msssim_finalize");
}

extern "C" int dsic_sqerr_per_image(const float* a, const float* b, double* out, int B,
                                    int64_t n_per_image, int clamp_a, void* stream) {
  DSIC_REQUIRE(a && b && out && B > 0 && n_per_image > 0, "sqerr: bad argument");
  hipLaunchKernelGGL(sqerr_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, a, b, out, n_per_image,
                     clamp_a);
  return check_launch("sqerr");
}
