// Distortion metrics on the GPU: MS-SSIM / SSIM and MSE.
//
// MS-SSIM restates pytorch-msssim 1.0.0 (called at modelseval.py:78-88 and
// eval_selfcontained_entropy.py:154; third-party, not vendored in the
// reference, so parity is pinned only by the repo's own oracle):
// 11-tap Gaussian window (sigma 1.5), separable "valid" filtering of
// X, Y, X*X, Y*Y, X*Y (along H first, then along W), C1=(0.01 L)^2, C2=(0.03 L)^2,
// cs = (2 s_xy + C2)/(s_x + s_y + C2), ssim = (2 m_x m_y + C1)/(m_x^2+m_y^2+C1)*cs,
// spatial means per (image, channel), 2x2 average pooling (padding = size%2,
// zeros counted) between levels, prod_l relu(v_l)^w_l, mean over channels.
//
// ssim_level_kernel streams: one WAVE owns a band of output rows of a 256-column strip (64 lanes x 4
// adjacent columns, 16-byte global loads; planes of <= 128 / <= 64 columns share a wave two / four at a
// time).  The filter along H runs out of an 11-row register ring (the loop is unrolled 11 times, so the
// ring slots are compile-time registers), the filter along W through a wave-private LDS row (one
// ds_write_b128 + four ds_read_b128 per map: 14 neighbours for 4 outputs) - no workgroup barrier
// anywhere.  The 2x2 average pool that feeds the next level is produced by the wave that owns the rows
// (even H, W % 4 == 0; other shapes use avgpool2_kernel).  Every output sums its 11 taps in ascending
// order; fixed-order fp64 partial sums (deterministic run to run).
#include "common.h"

namespace dsic {

#define WIN 11

struct GaussWin {
  float g[WIN];
};

struct SsimGeom {
  int segw, strips, rb, nbands, groups;
};
constexpr int SSIM_STRIP = 244;   // owned output columns per 256-column strip (256 - 10, rounded down to 4)

static SsimGeom ssim_geom(int planes, int H, int W) {
  SsimGeom g;
  const int Ho = H - (WIN - 1), Wo = W - (WIN - 1);
  g.segw = W <= 64 ? 64 : W <= 128 ? 128 : 256;
  g.strips = W <= 256 ? 1 : ceil_div(Wo, SSIM_STRIP);
  g.groups = ceil_div(planes, 256 / g.segw);
  // bands of output rows: about two waves per SIMD over the chip (2048 wave tasks), at least 16 rows each
  // (every band re-reads 10 halo rows), an even count so that the 2x2 pooling pairs stay inside a band
  // one round of waves: at most 2048 wave tasks (two per SIMD; more waves add nothing, the kernel is bound by vector
  // instruction issue), at least 8 rows each (every band re-reads 10 halo rows)
  const int maxb = 2048 / (g.groups * g.strips) > 0 ? 2048 / (g.groups * g.strips) : 1;
  int rb = ceil_div(Ho, maxb);
  if (rb < 8) rb = 8;
  rb += rb & 1;
  g.rb = rb;
  g.nbands = ceil_div(Ho, rb);
  return g;
}

template <int SEGW, bool VEC>
__global__ __launch_bounds__(256, 2) void ssim_level_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                         double* __restrict__ partial, float* __restrict__ Xn,
                                                         float* __restrict__ Yn, int planes, int H, int W, int rb,
                                                         int nbands, int strips, float C1, float C2, int clamp_x,
                                                         GaussWin gw) {
  typedef float floatx4 __attribute__((ext_vector_type(4)));
  typedef float floatx2 __attribute__((ext_vector_type(2)));
  constexpr int LPS = SEGW / 4;        // lanes per plane segment
  constexpr int NSEG = 64 / LPS;       // planes per wave
  constexpr int ROWF = 256 + 16;       // floats per LDS map row: 256 columns + the 14-wide reads of the last lanes
  __shared__ __attribute__((aligned(16))) float hbuf[4][5][ROWF];   // per wave: pair map (0,1) = rows 0-1, pair map (2,3) = rows 2-3, map 4 = row 4
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int groups = (planes + NSEG - 1) / NSEG;
  const int task = blockIdx.x * 4 + wave;
  if (task >= groups * strips * nbands) return;   // wave-uniform; the kernel has no barrier
  const int band = task % nbands;
  const int strip = (task / nbands) % strips;
  const int group = task / (nbands * strips);
  const int seg = lane / LPS, c4 = lane % LPS;
  const int plane = group * NSEG + seg;
  const bool pvalid = plane < planes;
  const int col0 = strip * SSIM_STRIP + 4 * c4;
  const int Ho = H - (WIN - 1), Wo = W - (WIN - 1);
  const int orow0 = band * rb;
  const int nout = min(rb, Ho - orow0);
  const int own_end = band == nbands - 1 ? H : orow0 + rb;           // input rows this wave pools
  const int col_end = strips == 1 ? Wo : min(Wo, (strip + 1) * SSIM_STRIP);   // output columns this wave owns
  const bool pool = Xn != nullptr;
  const int ocol_end = pvalid ? col_end : 0;   // no output column of an absent plane is counted
  const float* xp = X + (size_t)(pvalid ? plane : 0) * H * W;
  const float* yp = Y + (size_t)(pvalid ? plane : 0) * H * W;

  auto load_row = [&](int r, floatx4& a, floatx4& b) {
    a = floatx4{0.f, 0.f, 0.f, 0.f};
    b = a;
    if (VEC) {
      if (pvalid && col0 < W) {
        a = *(const floatx4*)(xp + (size_t)r * W + col0);
        b = *(const floatx4*)(yp + (size_t)r * W + col0);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (pvalid && col0 + e < W) {
          a[e] = xp[(size_t)r * W + col0 + e];
          b[e] = yp[(size_t)r * W + col0 + e];
        }
    }
    if (clamp_x) {
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] = fminf(fmaxf(a[e], 0.f), 1.f);
    }
  };
  // The five maps are filtered in packed pairs (v_pk_fma_f32: one instruction, two maps): (X, Y), (X*X, Y*Y), and
  // X*Y alone.  Ring slot = one input row as four (x, y) pairs.
  struct Row {
    floatx2 p[4];
  };
  auto pack_row = [&](const floatx4& a, const floatx4& b) {
    Row r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r.p[e] = floatx2{a[e], b[e]};
    return r;
  };
  // 2x2 average of rows (r-1, r), r odd: same summation order as avgpool2_kernel
  auto pool_rows = [&](int r, const Row& r0, const Row& r1) {
    if (!(pool && (r & 1) && r < own_end && pvalid)) return;
    if (col0 >= W || (strips > 1 && col0 >= (strip + 1) * SSIM_STRIP)) return;
    const size_t o = ((size_t)plane * (H >> 1) + (r >> 1)) * (W >> 1) + (col0 >> 1);
    // both planes at once: lane [0] = X, lane [1] = Y
    const floatx2 q0 = (((r0.p[0] + r0.p[1]) + r1.p[0]) + r1.p[1]) * 0.25f;
    const floatx2 q1 = (((r0.p[2] + r0.p[3]) + r1.p[2]) + r1.p[3]) * 0.25f;
    *(floatx2*)(Xn + o) = floatx2{q0[0], q1[0]};
    *(floatx2*)(Yn + o) = floatx2{q0[1], q1[1]};
  };

  Row ring[WIN];   // the 11 input rows under the current output row: slot k = tap k
#pragma unroll
  for (int i = 0; i < WIN - 1; ++i) {
    floatx4 a, b;
    load_row(orow0 + i, a, b);
    ring[i] = pack_row(a, b);
    if (i > 0) pool_rows(orow0 + i, ring[i - 1], ring[i]);
  }
  floatx4 pfa, pfb;
  load_row(orow0 + WIN - 1, pfa, pfb);
  double cs = 0.0, ss = 0.0;
  // LDS row of the wave: [pair map 0: 2 x ROWF][pair map 1: 2 x ROWF][map 4: ROWF]; a lane's four columns of a
  // pair map are 32 contiguous bytes
  float* const lds01 = &hbuf[wave][0][0] + 8 * lane;
  float* const lds23 = &hbuf[wave][2][0] + 8 * lane;
  float* const lds4 = &hbuf[wave][4][0] + 4 * lane;
  // One output row per iteration; the ring is shifted by register moves (tap k = slot k).  (Unrolling 11 rows with
  // rotating slot indices needs no moves but is > 100 KB of code: the instruction cache then bounds the kernel.)
#pragma nounroll
  for (int o = 0; o < nout; ++o) {
    {
      {
        constexpr int u = 0;
        const int r = orow0 + o + WIN - 1;
        ring[WIN - 1] = pack_row(pfa, pfb);
        if (o + 1 < nout) load_row(r + 1, pfa, pfb);
        pool_rows(r, ring[WIN - 2], ring[WIN - 1]);
        // filter along H, taps in ascending order, straight into the wave's LDS row
        {
          floatx2 m01[4], m23[4];
          floatx4 m4;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            m01[e] = floatx2{0.f, 0.f};
            m23[e] = floatx2{0.f, 0.f};
            m4[e] = 0.f;
          }
#pragma unroll
          for (int k = 0; k < WIN; ++k) {
            const floatx2 g2 = {gw.g[k], gw.g[k]};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const floatx2 v = ring[(u + k) % WIN].p[e];
              m01[e] = __builtin_elementwise_fma(g2, v, m01[e]);
              m23[e] = __builtin_elementwise_fma(g2, v * v, m23[e]);
              m4[e] = __builtin_fmaf(gw.g[k], v[0] * v[1], m4[e]);
            }
          }
          *(floatx4*)(lds01) = floatx4{m01[0][0], m01[0][1], m01[1][0], m01[1][1]};
          *(floatx4*)(lds01 + 4) = floatx4{m01[2][0], m01[2][1], m01[3][0], m01[3][1]};
          *(floatx4*)(lds23) = floatx4{m23[0][0], m23[0][1], m23[1][0], m23[1][1]};
          *(floatx4*)(lds23 + 4) = floatx4{m23[2][0], m23[2][1], m23[3][0], m23[3][1]};
          *(floatx4*)(lds4) = m4;
        }
        __builtin_amdgcn_sched_barrier(0);
        // filter along W through the wave's LDS row (in-order per wave: no barrier, only the data wait)
        floatx2 h01[4], h23[4];
        float h4[4];
        {
          floatx2 t[14];
#pragma unroll
          for (int v = 0; v < 7; ++v) {
            const floatx4 x = *(const floatx4*)(lds01 + 4 * v);
            t[2 * v] = floatx2{x[0], x[1]};
            t[2 * v + 1] = floatx2{x[2], x[3]};
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) h01[j] = floatx2{0.f, 0.f};
#pragma unroll
          for (int k = 0; k < WIN; ++k)   // the four outputs' chains side by side (a dependent v_pk_fma_f32 needs a wait state)
#pragma unroll
            for (int j = 0; j < 4; ++j) h01[j] = __builtin_elementwise_fma(floatx2{gw.g[k], gw.g[k]}, t[j + k], h01[j]);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
          floatx2 t[14];
#pragma unroll
          for (int v = 0; v < 7; ++v) {
            const floatx4 x = *(const floatx4*)(lds23 + 4 * v);
            t[2 * v] = floatx2{x[0], x[1]};
            t[2 * v + 1] = floatx2{x[2], x[3]};
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) h23[j] = floatx2{0.f, 0.f};
#pragma unroll
          for (int k = 0; k < WIN; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) h23[j] = __builtin_elementwise_fma(floatx2{gw.g[k], gw.g[k]}, t[j + k], h23[j]);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
          float t[16];
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const floatx4 x = *(const floatx4*)(lds4 + 4 * v);
            t[4 * v] = x[0]; t[4 * v + 1] = x[1]; t[4 * v + 2] = x[2]; t[4 * v + 3] = x[3];
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) h4[j] = 0.f;
#pragma unroll
          for (int k = 0; k < WIN; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) h4[j] = __builtin_fmaf(gw.g[k], t[j + k], h4[j]);
        }
        __builtin_amdgcn_sched_barrier(0);
        float rcs[4], rss[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // computed by every lane (no branch), counted by the lanes that own the output
          const floatx2 musq = h01[j] * h01[j];            // (mu1^2, mu2^2)
          const float mu12 = h01[j][0] * h01[j][1];
          const floatx2 sg = h23[j] - musq;                // (sigma1^2, sigma2^2)
          const float s12 = h4[j] - mu12;
          // a * rcp(b) (v_rcp_f32, 1 ulp) instead of the IEEE division sequence (11 instructions each, 8 per row):
          // 1e-7 relative per pixel against a 1e-4 bar on the mean
          const float csv = (2.f * s12 + C2) * __builtin_amdgcn_rcpf(sg[0] + sg[1] + C2);
          const float sv = ((2.f * mu12 + C1) * __builtin_amdgcn_rcpf(musq[0] + musq[1] + C1)) * csv;
          const bool ok = col0 + j < ocol_end;
          rcs[j] = ok ? csv : 0.f;
          rss[j] = ok ? sv : 0.f;
        }
        // the row's four values in fp32 (fixed order), the running sums in fp64
        cs += (double)(((rcs[0] + rcs[1]) + rcs[2]) + rcs[3]);
        ss += (double)(((rss[0] + rss[1]) + rss[2]) + rss[3]);
#pragma unroll
        for (int i = 0; i < WIN - 1; ++i) ring[i] = ring[i + 1];
      }
    }
  }
#pragma unroll
  for (int o = LPS / 2; o > 0; o >>= 1) {
    cs += __shfl_down(cs, o, LPS);
    ss += __shfl_down(ss, o, LPS);
  }
  if (c4 == 0 && pvalid) {
    const size_t o = (((size_t)plane * strips + strip) * nbands + band) * 2;
    partial[o] = cs;
    partial[o + 1] = ss;
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

static int ssim_level_launch(const float* X, const float* Y, double* partial, double* means, float* Xn, float* Yn,
                             int planes, int H, int W, float C1, float C2, int clamp_x, hipStream_t st) {
  const int Ho = H - (WIN - 1), Wo = W - (WIN - 1);
  const SsimGeom g = ssim_geom(planes, H, W);
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
  const int ntasks = g.groups * g.strips * g.nbands;
  const dim3 grid((unsigned)ceil_div(ntasks, 4)), block(256);
  const bool vec = W % 4 == 0 && ((uintptr_t)X % 16) == 0 && ((uintptr_t)Y % 16) == 0;
#define SSIM_LAUNCH(S, V)                                                                                           \
  hipLaunchKernelGGL((ssim_level_kernel<S, V>), grid, block, 0, st, X, Y, partial, Xn, Yn, planes, H, W, g.rb, g.nbands, \
                     g.strips, C1, C2, clamp_x, gw)
  if (g.segw == 64) { if (vec) SSIM_LAUNCH(64, true); else SSIM_LAUNCH(64, false); }
  else if (g.segw == 128) { if (vec) SSIM_LAUNCH(128, true); else SSIM_LAUNCH(128, false); }
  else { if (vec) SSIM_LAUNCH(256, true); else SSIM_LAUNCH(256, false); }
#undef SSIM_LAUNCH
  int rc = check_launch("ssim_level");
  if (rc) return rc;
  hipLaunchKernelGGL(ssim_reduce_kernel, dim3(ceil_div(planes, 64)), dim3(64), 0, st, partial, means,
                     g.strips * g.nbands, 1.0 / ((double)Ho * Wo), planes);
  return check_launch("ssim_reduce");
}

extern "C" int dsic_ssim_level(const float* X, const float* Y, double* partial, double* means,
                               int planes, int H, int W, float C1, float C2, int clamp_x,
                               void* stream) {
  DSIC_REQUIRE(X && Y && partial && means, "ssim_level: null pointer");
  DSIC_REQUIRE(planes > 0 && H >= WIN && W >= WIN, "ssim_level: plane %dx%d smaller than the 11x11 window", H, W);
  return ssim_level_launch(X, Y, partial, means, nullptr, nullptr, planes, H, W, C1, C2, clamp_x, (hipStream_t)stream);
}

extern "C" int dsic_ssim_level_pool_fused(int H, int W) { return H % 2 == 0 && W % 4 == 0; }

extern "C" int dsic_ssim_level_pool(const float* X, const float* Y, double* partial, double* means, float* Xn,
                                    float* Yn, int planes, int H, int W, float C1, float C2, int clamp_x,
                                    void* stream) {
  DSIC_REQUIRE(X && Y && partial && means && Xn && Yn, "ssim_level_pool: null pointer");
  DSIC_REQUIRE(planes > 0 && H >= WIN && W >= WIN, "ssim_level_pool: plane %dx%d smaller than the 11x11 window", H, W);
  if (dsic_ssim_level_pool_fused(H, W) && (uintptr_t)X % 16 == 0 && (uintptr_t)Y % 16 == 0 && (uintptr_t)Xn % 8 == 0 &&
      (uintptr_t)Yn % 8 == 0)
    return ssim_level_launch(X, Y, partial, means, Xn, Yn, planes, H, W, C1, C2, clamp_x, (hipStream_t)stream);
  // odd sizes (padded pooling): the level and the two pools as separate passes
  int rc = ssim_level_launch(X, Y, partial, means, nullptr, nullptr, planes, H, W, C1, C2, clamp_x, (hipStream_t)stream);
  if (rc) return rc;
  rc = dsic_avgpool2(X, Xn, planes, H, W, clamp_x, stream);
  if (rc) return rc;
  return dsic_avgpool2(Y, Yn, planes, H, W, 0, stream);
}

extern "C" int64_t dsic_ssim_partial_doubles(int planes, int H, int W) {
  if (H < WIN || W < WIN || planes <= 0) return 0;
  const SsimGeom g = ssim_geom(planes, H, W);
  return (int64_t)planes * g.strips * g.nbands * 2;
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
