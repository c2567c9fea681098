// Small-tensor tail of the encoder: hyper-synthesis heads, quantisation and the
// Student-t / Gaussian rate terms.  HBM-bound scans with wave-shuffle
// reductions; one workgroup per image so every per-image sum has a fixed,
// run-to-run deterministic reduction tree.
//
//   dsic_hyper_params : AdaptiveAvgPool2d(1) + mlp_sigma/mlp_nu (layers.py:131-139,
//                       147-151) + exp / clamp (model.py:54-55)
//   dsic_rate         : quantize (model.py:27-35), StudentT.neg_log2_prob
//                       (distributions.py:20-31), FactorizedGaussian.neg_log2_prob
//                       (distributions.py:39-46), per-image sums (model.py:76)
//   dsic_gdn_nchw     : stand-alone GDN/IGDN (layers.py:19-27)
#include "common.h"

namespace dsic {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// block-wide double sum; result valid in thread 0. scratch: >= 4 doubles of LDS.
__device__ __forceinline__ double block_sum(double v, double* scratch) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0)
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r += scratch[w];
  return r;
}

#define MAXN 256

__global__ __launch_bounds__(256) void hyper_params_kernel(
    const float* __restrict__ t, const float* __restrict__ w1s, const float* __restrict__ b1s,
    const float* __restrict__ w2s, const float* __restrict__ b2s, const float* __restrict__ w1n,
    const float* __restrict__ b1n, const float* __restrict__ w2n, const float* __restrict__ b2n,
    float* __restrict__ log_sigma, float* __restrict__ log_nu, float* __restrict__ sigma,
    float* __restrict__ nu, int HW, int N, int M, float min_nu, float max_nu) {
  __shared__ float part[2][MAXN];
  __shared__ float pooled[MAXN];
  __shared__ float hid[2][MAXN];
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const float* tb = t + (size_t)b * HW * N;
  // pool: thread (half, c) sums every second pixel of channel c; N <= 128 uses
  // both halves, N in (128,256] one pixel stream per thread.
  const int halves = N <= 128 ? 2 : 1;
  const int c = tid % (N <= 128 ? 128 : 256);
  const int half = N <= 128 ? tid / 128 : 0;
  float s = 0.f;
  if (c < N)
    for (int p = half; p < HW; p += halves) s += tb[(size_t)p * N + c];
  if (c < N) part[half][c] = s;
  __syncthreads();
  if (tid < N) {
    float v = part[0][tid];
    if (halves == 2) v += part[1][tid];
    pooled[tid] = v / (float)HW;
  }
  __syncthreads();
  // hidden layers: weights stored input-major [N_in][N_out] so lanes read coalesced
  for (int o = tid; o < 2 * N; o += 256) {
    const int head = o / N, oo = o % N;
    const float* w1 = head ? w1n : w1s;
    float a = head ? b1n[oo] : b1s[oo];
    for (int i = 0; i < N; ++i) a += w1[(size_t)i * N + oo] * pooled[i];
    hid[head][oo] = a > 0.f ? a : 0.f;
  }
  __syncthreads();
  for (int o = tid; o < 2 * M; o += 256) {
    const int head = o / M, m = o % M;
    const float* w2 = head ? w2n : w2s;
    float a = head ? b2n[m] : b2s[m];
    for (int i = 0; i < N; ++i) a += w2[(size_t)i * M + m] * hid[head][i];
    if (head == 0) {
      log_sigma[(size_t)b * M + m] = a;
      sigma[(size_t)b * M + m] = expf(a);
    } else {
      log_nu[(size_t)b * M + m] = a;
      nu[(size_t)b * M + m] = fminf(fmaxf(expf(a), min_nu), max_nu);
    }
  }
}

#define MAXM 512
#define LOG2E_F 1.4426950408889634f
#define TWO_PI_F 6.283185307179586f
#define PI_D 3.141592653589793

#define RATE_SPLIT 16  // workgroups per image
#define RATE_PT 16     // pixels per LDS tile

__global__ __launch_bounds__(256) void rate_kernel(
    const float* __restrict__ y, const float* __restrict__ z, const float* __restrict__ y_noisy,
    const float* __restrict__ z_noisy, const float* __restrict__ sigma, const float* __restrict__ nu,
    const float* __restrict__ z_log_sigma, float* __restrict__ y_hat, float* __restrict__ y_tilde,
    float* __restrict__ z_tilde, float* __restrict__ nll_y, float* __restrict__ nll_z,
    double* __restrict__ part, int HWy, int M, int HWz, int N, int per_element) {
  // grid (RATE_SPLIT, B): block (s, b) takes the s-th range of image b's y pixels (a multiple of 4 pixels, all
  // channels); the last one also takes z.  part[b][s][2] = this block's {sum nll_y, sum nll_z};
  // rate_reduce_kernel adds the slices in fixed order (deterministic, independent of the batch).
  // The NCHW outputs (y_tilde, nll_y) leave through an LDS tile of RATE_PT pixels x M channels: 16-byte stores of
  // 4 consecutive pixels of a channel instead of 4-byte stores 4 HWy bytes apart (165 MB written per launch for
  // 38 MB of output in round 2).
  __shared__ float s_sig[MAXM], s_nu[MAXM], s_logc[MAXM];
  __shared__ double scratch[4];
  extern __shared__ __attribute__((aligned(16))) float rate_tile[];   // [2][M][RATE_PT + 1]
  float* const t_x = rate_tile;
  float* const t_b = rate_tile + (size_t)M * (RATE_PT + 1);
  const int b = blockIdx.y, slice = blockIdx.x, tid = threadIdx.x;
  const int pps = ((HWy + RATE_SPLIT - 1) / RATE_SPLIT + 3) & ~3;   // pixels per slice
  const int p_begin = min(HWy, slice * pps), p_end = min(HWy, p_begin + pps);
  for (int c = tid; c < M && !per_element; c += 256) {
    // distributions.py:25-29; the channel constant is evaluated in fp64 and
    // rounded once (the reference evaluates it in fp32 per element).
    const float sg = fminf(fmaxf(sigma[(size_t)b * M + c], 1e-3f), 1e3f);
    const float nv = fminf(fmaxf(nu[(size_t)b * M + c], 2.0f), 100.0f);
    const double nd = (double)nv;
    const double lc = lgamma((nd + 1.0) * 0.5) - lgamma(nd * 0.5) - 0.5 * log(nd * PI_D) - log((double)sg);
    s_sig[c] = sg;
    s_nu[c] = nv;
    s_logc[c] = (float)lc;
  }
  __syncthreads();
  double acc = 0.0;
  const size_t ybase = (size_t)b * HWy * M;
  const bool vec = (HWy & 3) == 0;
  for (int p0 = p_begin; p0 < p_end; p0 += RATE_PT) {
    const int npx = min(RATE_PT, p_end - p0);
    for (int i = tid; i < npx * M; i += 256) {  // i walks NHWC (coalesced reads)
      const int pl = i / M, c = i - pl * M;
      const int p = p0 + pl;
      const size_t g = ybase + (size_t)p * M + c;
      const float v = y[g];
      const float r = rintf(v);  // torch.round: half to even
      const float xt = y_noisy ? y_noisy[g] : r;
      float sg, nv, lc;
      if (per_element) {  // spatial_params (model.py:49-51): sigma, nu are NCHW [B,M,HWy]
        const size_t e = ((size_t)b * M + c) * HWy + p;
        sg = fminf(fmaxf(sigma[e], 1e-3f), 1e3f);
        nv = fminf(fmaxf(nu[e], 2.0f), 100.0f);
        const double nd = (double)nv;
        lc = (float)(lgamma((nd + 1.0) * 0.5) - lgamma(nd * 0.5) - 0.5 * log(nd * PI_D) - log((double)sg));
      } else {
        sg = s_sig[c];
        nv = s_nu[c];
        lc = s_logc[c];
      }
      const float q = xt / sg;
      const float quad = q * q;
      const float logp = lc - ((nv + 1.0f) / 2.0f) * log1pf(quad / nv);
      const float bits = -logp * LOG2E_F;
      y_hat[g] = r;
      t_x[c * (RATE_PT + 1) + pl] = xt;
      t_b[c * (RATE_PT + 1) + pl] = bits;
      acc += (double)bits;
    }
    __syncthreads();
    for (int i = tid; i < M * (RATE_PT / 4); i += 256) {
      const int c = i / (RATE_PT / 4), g4 = i - c * (RATE_PT / 4);
      const int pl = 4 * g4;
      if (pl >= npx) continue;
      const float* tx = t_x + c * (RATE_PT + 1) + pl;
      const float* tb = t_b + c * (RATE_PT + 1) + pl;
      const size_t o = ((size_t)b * M + c) * HWy + p0 + pl;
      if (vec && pl + 3 < npx) {
        *(float4*)(y_tilde + o) = make_float4(tx[0], tx[1], tx[2], tx[3]);
        *(float4*)(nll_y + o) = make_float4(tb[0], tb[1], tb[2], tb[3]);
      } else {
        for (int e = 0; e < 4 && pl + e < npx; ++e) {
          y_tilde[o + e] = tx[e];
          nll_y[o + e] = tb[e];
        }
      }
    }
    __syncthreads();
  }
  const double sy = block_sum(acc, scratch);
  acc = 0.0;
  const size_t zbase = (size_t)b * HWz * N;
  for (int i = tid; i < (slice == RATE_SPLIT - 1 ? HWz * N : 0); i += 256) {
    const int p = i / N, c = i % N;
    const float sg = fminf(fmaxf(expf(z_log_sigma[c]), 1e-3f), 1e3f);
    const float var = sg * sg;
    const float xt = z_noisy ? z_noisy[zbase + i] : rintf(z[zbase + i]);
    const float logp = -0.5f * logf(TWO_PI_F * var) - 0.5f * (xt * xt) / var;
    const float bits = -logp * LOG2E_F;
    const size_t o = ((size_t)b * N + c) * HWz + p;
    z_tilde[o] = xt;
    nll_z[o] = bits;
    acc += (double)bits;
  }
  const double sz = block_sum(acc, scratch);
  if (tid == 0) {
    part[((size_t)b * RATE_SPLIT + slice) * 2] = sy;
    part[((size_t)b * RATE_SPLIT + slice) * 2 + 1] = sz;
  }
}

__global__ void rate_reduce_kernel(const double* __restrict__ part, double* __restrict__ sums, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double sy = 0.0, sz = 0.0;
  for (int s = 0; s < RATE_SPLIT; ++s) {
    sy += part[((size_t)b * RATE_SPLIT + s) * 2];
    sz += part[((size_t)b * RATE_SPLIT + s) * 2 + 1];
  }
  sums[2 * b] = sy;
  sums[2 * b + 1] = sz;
}

__global__ void gdn_nchw_kernel(const float* __restrict__ x, const float* __restrict__ beta,
                                const float* __restrict__ gamma, float* __restrict__ out, int C,
                                int HW, int inverse, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (i / HW) % C;
  const float v = x[i];
  const float d = __fsqrt_rn(__fadd_rn(beta[c], __fmul_rn(gamma[c], __fmul_rn(v, v))));
  out[i] = inverse ? __fmul_rn(v, d) : __fdiv_rn(v, d);
}

// Stand-alone prior evaluations for callers that use the distribution classes
// directly (same arithmetic as rate_kernel).
__global__ void student_t_bits_kernel(const float* __restrict__ x, const float* __restrict__ sigma,
                                      const float* __restrict__ nu, float* __restrict__ out, int HW,
                                      int per_channel, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int64_t pi = per_channel ? i / HW : i;
  const float sg = fminf(fmaxf(sigma[pi], 1e-3f), 1e3f);
  const float nv = fminf(fmaxf(nu[pi], 2.0f), 100.0f);
  const double nd = (double)nv;
  const float logc =
      (float)(lgamma((nd + 1.0) * 0.5) - lgamma(nd * 0.5) - 0.5 * log(nd * PI_D) - log((double)sg));
  const float q = x[i] / sg;
  const float logp = logc - ((nv + 1.0f) / 2.0f) * log1pf((q * q) / nv);
  out[i] = -logp * LOG2E_F;
}

__global__ void gaussian_bits_kernel(const float* __restrict__ x, const float* __restrict__ log_sigma,
                                     float* __restrict__ out, int C, int HW, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (i / HW) % C;
  const float sg = fminf(fmaxf(expf(log_sigma[c]), 1e-3f), 1e3f);
  const float var = sg * sg;
  const float v = x[i];
  const float logp = -0.5f * logf(TWO_PI_F * var) - 0.5f * (v * v) / var;
  out[i] = -logp * LOG2E_F;
}

// spatial_params branch of model.py:49-51: sigma = exp(log_sigma), nu = clamp(exp(log_nu)),
// from the NHWC head outputs to the reference's NCHW tensors.
__global__ void sigma_nu_spatial_kernel(const float* __restrict__ ls, const float* __restrict__ ln,
                                        float* __restrict__ sigma, float* __restrict__ nu, int HW, int M,
                                        float min_nu, float max_nu, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // NCHW index
  if (i >= total) return;
  const int p = i % HW;
  const int c = (i / HW) % M;
  const int64_t b = i / ((int64_t)HW * M);
  const size_t src = ((size_t)b * HW + p) * M + c;
  sigma[i] = expf(ls[src]);
  nu[i] = fminf(fmaxf(expf(ln[src]), min_nu), max_nu);
}

__global__ void round_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) out[i] = rintf(x[i]);
}

}  // namespace dsic

using namespace dsic;

extern "C" int dsic_student_t_bits(const float* x, const float* sigma, const float* nu, float* out,
                                   int64_t n, int HW, int per_channel, void* stream) {
  DSIC_REQUIRE(x && sigma && nu && out && n > 0 && HW > 0, "student_t_bits: bad argument");
  hipLaunchKernelGGL(student_t_bits_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, x, sigma, nu, out, HW, per_channel, n);
  return check_launch("student_t_bits");
}

extern "C" int dsic_gaussian_bits(const float* x, const float* log_sigma, float* out, int B, int C,
                                  int HW, void* stream) {
  DSIC_REQUIRE(x && log_sigma && out && B > 0 && C > 0 && HW > 0, "gaussian_bits: bad argument");
  const int64_t n = (int64_t)B * C * HW;
  hipLaunchKernelGGL(gaussian_bits_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, x, log_sigma, out, C, HW, n);
  return check_launch("gaussian_bits");
}

extern "C" int dsic_sigma_nu_spatial(const float* log_sigma_nhwc, const float* log_nu_nhwc, float* sigma_nchw,
                                     float* nu_nchw, int B, int HW, int M, float min_nu, float max_nu,
                                     void* stream) {
  DSIC_REQUIRE(log_sigma_nhwc && log_nu_nhwc && sigma_nchw && nu_nchw && B > 0 && HW > 0 && M > 0,
               "sigma_nu_spatial: bad argument");
  const int64_t total = (int64_t)B * HW * M;
  hipLaunchKernelGGL(sigma_nu_spatial_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, log_sigma_nhwc, log_nu_nhwc, sigma_nchw, nu_nchw, HW, M, min_nu, max_nu,
                     total);
  return check_launch("sigma_nu_spatial");
}

extern "C" int dsic_round(const float* x, float* out, int64_t n, void* stream) {
  DSIC_REQUIRE(x && out && n > 0, "round: bad argument");
  hipLaunchKernelGGL(round_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     x, out, n);
  return check_launch("round");
}

extern "C" int dsic_hyper_params(const float* t_nhwc, const float* w1_sigma, const float* b1_sigma,
                                 const float* w2_sigma, const float* b2_sigma, const float* w1_nu,
                                 const float* b1_nu, const float* w2_nu, const float* b2_nu,
                                 float* log_sigma, float* log_nu, float* sigma, float* nu, int B,
                                 int HW, int N, int M, float min_nu, float max_nu, void* stream) {
  DSIC_REQUIRE(t_nhwc && w1_sigma && b1_sigma && w2_sigma && b2_sigma && w1_nu && b1_nu && w2_nu &&
                   b2_nu && log_sigma && log_nu && sigma && nu, "hyper_params: null pointer");
  DSIC_REQUIRE(B > 0 && HW > 0, "hyper_params: empty tensor");
  DSIC_REQUIRE(N > 0 && N <= MAXN && M > 0, "hyper_params: N=%d must be in [1,%d]", N, MAXN);
  hipLaunchKernelGGL(hyper_params_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, t_nhwc, w1_sigma,
                     b1_sigma, w2_sigma, b2_sigma, w1_nu, b1_nu, w2_nu, b2_nu, log_sigma, log_nu, sigma,
                     nu, HW, N, M, min_nu, max_nu);
  return check_launch("hyper_params");
}

extern "C" int dsic_rate(const float* y_nhwc, const float* z_nhwc, const float* y_noisy_nhwc,
                         const float* z_noisy_nhwc, const float* sigma, const float* nu,
                         const float* z_log_sigma, float* y_hat_nhwc, float* y_tilde_nchw,
                         float* z_tilde_nchw, float* nll_y_nchw, float* nll_z_nchw, double* sums,
                         double* work, int B, int HWy, int M, int HWz, int N, int per_element,
                         void* stream) {
  DSIC_REQUIRE(y_nhwc && z_nhwc && sigma && nu && z_log_sigma && y_hat_nhwc && y_tilde_nchw &&
                   z_tilde_nchw && nll_y_nchw && nll_z_nchw && sums && work, "rate: null pointer");
  DSIC_REQUIRE(B > 0 && HWy > 0 && HWz > 0, "rate: empty tensor");
  DSIC_REQUIRE(M > 0 && M <= MAXM && N > 0, "rate: M=%d must be in [1,%d]", M, MAXM);
  DSIC_REQUIRE((int64_t)HWy * M < ((int64_t)1 << 31), "rate: latent too large");
  DSIC_REQUIRE(B <= 65535, "rate: B=%d exceeds the grid limit", B);
  const size_t tile_bytes = (size_t)2 * M * (RATE_PT + 1) * sizeof(float);   // <= 69 632 (M = 512)
  if (tile_bytes > 40000) {   // beside the 6 KB of static LDS: ask for more than the default 64 KB
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    static bool attr_set[64] = {};
    if (!attr_set[dev]) {
      const hipError_t e = hipFuncSetAttribute((const void*)rate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               2 * MAXM * (RATE_PT + 1) * (int)sizeof(float));
      if (e != hipSuccess) {
        set_error("rate: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return DSIC_EHIP;
      }
      attr_set[dev] = true;
    }
  }
  hipLaunchKernelGGL(rate_kernel, dim3(RATE_SPLIT, B), dim3(256), tile_bytes, (hipStream_t)stream, y_nhwc, z_nhwc,
                     y_noisy_nhwc, z_noisy_nhwc, sigma, nu, z_log_sigma, y_hat_nhwc, y_tilde_nchw,
                     z_tilde_nchw, nll_y_nchw, nll_z_nchw, work, HWy, M, HWz, N, per_element ? 1 : 0);
  int rc = check_launch("rate");
  if (rc) return rc;
  hipLaunchKernelGGL(rate_reduce_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, (hipStream_t)stream, work, sums, B);
  return check_launch("rate_reduce");
}

extern "C" int64_t dsic_rate_workspace_doubles(int B) { return (int64_t)B * RATE_SPLIT * 2; }

extern "C" int dsic_gdn_nchw(const float* x, const float* beta, const float* gamma, float* out, int B,
                             int C, int HW, int inverse, void* stream) {
  DSIC_REQUIRE(x && beta && gamma && out, "gdn: null pointer");
  DSIC_REQUIRE(B > 0 && C > 0 && HW > 0, "gdn: empty tensor");
  const int64_t total = (int64_t)B * C * HW;
  hipLaunchKernelGGL(gdn_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, x, beta, gamma, out, C, HW, inverse, total);
  return check_launch("gdn");
}
