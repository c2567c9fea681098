// fp32 implicit-GEMM convolution for gfx950 (MI355X) on v_mfma_f32_32x32x2_f32.
//
// Replaces nn.Conv2d / nn.ConvTranspose2d + GDN/IGDN/ReLU of the reference's
// transforms (code/modelv2/layers.py:29-31, 46-152).  Exact-fp32 MFMA is used
// because the parity bar (|dbpp|,|dMS-SSIM| <= 1e-4 with round-to-even latents)
// does not survive bf16 operands (SURVEY.md §7 "Hard parts").
//
// GEMM view, per workgroup (256 threads = 4 waves, one per SIMD):
//   M = 128 output pixels  (TN images x TOH rows x TOW cols of the output grid)
//   N = all output channels (each wave owns 32-column tiles)
//   K = taps x Cin, walked as  for Cin-chunk (CK ch) { for tap { for 8-ch sub } }
// A (im2col rows) is never materialised: the input window of the tile
// (halo included, zero filled outside the image) is staged NHWC in LDS once per
// Cin-chunk and every tap reads it at a different pixel offset.
// B (weights) comes pre-packed [tap][Cin/8][CoutP][8] so that one
// global_load_dwordx4 per lane yields four k-steps; it is read straight from
// L2 into registers, software-pipelined one step ahead (1.6 MB per 5x5 layer is
// L2-resident and shared by every workgroup).
//
// MFMA operand map (32x32x2, lane l: r = l&31, h = l>>5): A[i=r][k=h], B[k=h][j=r].
// K order is free as long as A and B agree, so k-step s of an 8-channel
// sub-chunk pairs channel s (h=0) with channel 4+s (h=1): both fragments are
// then 16 contiguous bytes per lane (ds_read_b128 / global_load_dwordx4).
// LDS pixel stride is CK+4 dwords (4*odd mod 64): conflict-free ds_read_b128 for
// the stride-1 window walk.
//
// Transposed 5x5/s2 convolutions run as their four sub-pixel phases
// (blockIdx.y): phase (py,px) is a stride-1 conv over the input grid with
// (3-py)x(3-px) taps whose outputs land on (2*oy+py, 2*ox+px): no zero MACs.
#include <stdlib.h>

#include "common.h"

// -DDSIC_DIAG=1 builds the ablation switches read from the DSIC_DBG environment variable
// (tools/conv_bench.py); production builds compile them out.
#ifndef DSIC_DIAG
#define DSIC_DIAG 0
#endif
#define DIAG(bit) (DSIC_DIAG && (a.dbg & (bit)))

namespace dsic {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
  const float* in;
  const float* w;
  const float* bias;
  const float* beta;
  const float* gamma;
  float* out;
  int B, H, W, Cin;   // input tensor NHWC
  int Ho, Wo;         // output grid walked by the tiles (per phase for convT)
  int oH, oW;         // output tensor spatial dims
  int Cout, CoutP;    // stored output channels / padded columns of w
  int os;             // output stride: 1 conv, 2 convT phases
  int transposed;     // 1: four sub-pixel phases on blockIdx.y
  int act;
  int tiles_x, tiles_y;
  int dbg;            // diagnostic ablation bits (0 in production)
};

__device__ __forceinline__ float apply_act(float v, int act, float beta, float gamma) {
  if (act == DSIC_ACT_GDN || act == DSIC_ACT_IGDN) return gdn_apply(v, beta, gamma, act == DSIC_ACT_IGDN);
  if (act == DSIC_ACT_RELU) return v > 0.f ? v : 0.f;
  return v;
}

// WIN: window extent per dim (3 or 5); S: input stride; TOW/TOH/TN: tile shape
// (product 128); CK: channels staged per chunk; NTW: 32-column tiles per wave;
// NARROW: CoutP == 32, waves split the M tiles instead of the N tiles.
template <int WIN, int S, int TOW, int TOH, int TN, int CK, int NTW, bool NARROW>
__global__ __launch_bounds__(256, NTW == 2 ? 1 : 2) void conv_igemm_kernel(const ConvArgs a) {
  static_assert(TOW * TOH * TN == 128, "tile must hold 128 output pixels");
  constexpr int TWIN = (TOW - 1) * S + WIN;
  constexpr int THIN = (TOH - 1) * S + WIN;
  constexpr int NPIX = TN * THIN * TWIN;
  constexpr int P = CK + 4;  // LDS dwords per pixel
  constexpr int Q = CK / 4;  // float4 slots per pixel
  constexpr int NSLOT = (NPIX * Q + 255) / 256;
  constexpr int NSUB = CK / 8;
  constexpr int MTW = NARROW ? 1 : 4;  // M tiles per wave
  constexpr int PAD = (WIN - 1) / 2;

  constexpr int EPI_STRIDE = 36;                    // floats per row of the epilogue transpose tile
  constexpr int EPI_FLOATS = 4 * 32 * EPI_STRIDE;   // one 32x32 tile per wave
  constexpr int LDS_FLOATS = NPIX * P > EPI_FLOATS ? NPIX * P : EPI_FLOATS;
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5;
  const int l31 = lane & 31;

  int bx = blockIdx.x;
  const int tile_x = bx % a.tiles_x;
  bx /= a.tiles_x;
  const int tile_y = bx % a.tiles_y;
  const int tile_n = bx / a.tiles_y;
  const int ox0 = tile_x * TOW, oy0 = tile_y * TOH, n0 = tile_n * TN;

  const int phase = blockIdx.y;
  const int py = phase >> 1, px = phase & 1;
  int nty = WIN, ntx = WIN, wy0 = 0, wx0 = 0, tapbase = 0;
  if (a.transposed) {
    nty = 3 - py;
    ntx = 3 - px;
    wy0 = py;
    wx0 = px;
    tapbase = phase == 0 ? 0 : (phase == 1 ? 9 : (phase == 2 ? 15 : 21));
  }
  const int Cin = a.Cin;
  const int Cin8 = Cin >> 3;
  const int nchunks = Cin / CK;

  // ---- per-thread staging slots: global element offsets (or -1) ------------
  const float* in_tile = a.in + (size_t)n0 * a.H * a.W * Cin;
  int goff[NSLOT];
  int loff[NSLOT];
#pragma unroll
  for (int i = 0; i < NSLOT; ++i) {
    const int slot = tid + i * 256;
    int g = -1, lo = -1;
    if (slot < NPIX * Q) {
      const int pix = slot / Q, q = slot % Q;
      const int wx = pix % TWIN;
      const int wy = (pix / TWIN) % THIN;
      const int tn = pix / (TWIN * THIN);
      const int gy = oy0 * S - PAD + wy;
      const int gx = ox0 * S - PAD + wx;
      lo = pix * P + q * 4;
      if (n0 + tn < a.B && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
        g = ((tn * a.H + gy) * a.W + gx) * Cin + q * 4;
    }
    goff[i] = g;
    loff[i] = lo;
  }

  // ---- A fragment bases (LDS dword offsets) --------------------------------
  int abase[MTW];
#pragma unroll
  for (int m = 0; m < MTW; ++m) {
    const int mt = NARROW ? wave : m;
    const int r = mt * 32 + l31;
    const int ox = r % TOW;
    const int oy = (r / TOW) % TOH;
    const int tn = r / (TOW * TOH);
    abase[m] = ((tn * THIN + oy * S) * TWIN + ox * S) * P + 4 * h;
  }

  // ---- B fragment lane offsets ---------------------------------------------
  // A wave whose column tile lies beyond CoutP recomputes tile 0 (in-bounds
  // weights) and drops the result in the epilogue: keeps the MFMA loop branch-free.
  int ntile[NTW];
  bool nvalid[NTW];
  int boff[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    ntile[j] = NARROW ? 0 : (int)blockIdx.z * 4 * NTW + wave + 4 * j;  // blockIdx.z: column block
    nvalid[j] = ntile[j] * 32 < a.CoutP;
    boff[j] = ((nvalid[j] ? ntile[j] : 0) * 32 + l31) * 8 + 4 * h;
  }
  const int wstep = a.CoutP * 8;  // floats per (tap, 8-channel) slab

  floatx16 acc[MTW][NTW];
#pragma unroll
  for (int m = 0; m < MTW; ++m)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[m][j][e] = 0.f;

  floatx4 stage[NSLOT];
  auto issue_chunk = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
      floatx4 v = {0.f, 0.f, 0.f, 0.f};
      if (goff[i] >= 0 && !DIAG(1)) v = *(const floatx4*)(in_tile + goff[i] + chunk * CK);
      stage[i] = v;
    }
  };

  issue_chunk(0);

  for (int chunk = 0; chunk < nchunks; ++chunk) {
#pragma unroll
    for (int i = 0; i < NSLOT; ++i)
      if (loff[i] >= 0 && !DIAG(2)) *(floatx4*)(lds + loff[i]) = stage[i];
    __syncthreads();
    if (chunk + 1 < nchunks) issue_chunk(chunk + 1);

    // weight slab walk for this chunk: slab(t, sub) = (tapbase+t)*Cin8 + chunk*NSUB + sub
    const float* wchunk = a.w + (size_t)(tapbase * Cin8 + chunk * NSUB) * wstep;
    const int nsteps = nty * ntx * NSUB;

    // Two operand sets in ping-pong: the set of step k+1 is loaded (LDS for A,
    // L2 for B) before the 16 MFMAs of step k issue, so no load latency sits
    // between MFMA clusters and no register copies are needed.
    floatx4 a0[MTW], a1[MTW], b0[NTW], b1[NTW];
    int ty = 0, tx = 0, sub = 0;  // cursor of the step whose operands are loaded next
    auto advance = [&]() {
      sub++;
      if (sub == NSUB) {
        sub = 0;
        tx++;
        if (tx == ntx) {
          tx = 0;
          ty++;
        }
      }
    };
    auto load_ops = [&](floatx4(&A)[MTW], floatx4(&Bf)[NTW]) {
      const float* wn = wchunk + (size_t)((ty * ntx + tx) * Cin8 + sub) * wstep;
#pragma unroll
      for (int j = 0; j < NTW; ++j)
        if (!DIAG(4)) Bf[j] = *(const floatx4*)(wn + boff[j]);
      const int aoff = ((ty + wy0) * TWIN + (tx + wx0)) * P + sub * 8;
#pragma unroll
      for (int m = 0; m < MTW; ++m) A[m] = *(const floatx4*)(lds + abase[m] + aoff);
    };
    auto mfma16 = [&](const floatx4(&A)[MTW], const floatx4(&Bf)[NTW]) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
          for (int m = 0; m < MTW; ++m)
            acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[m][s4], Bf[j][s4], acc[m][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    };
#pragma unroll
    for (int j = 0; j < NTW; ++j) b0[j] = b1[j] = floatx4{0.f, 0.f, 0.f, 0.f};
    load_ops(a0, b0);
    int step = 0;
    for (; step + 2 <= nsteps; step += 2) {
      advance();
      load_ops(a1, b1);
      mfma16(a0, b0);
      advance();
      if (step + 2 < nsteps) load_ops(a0, b0);
      mfma16(a1, b1);
    }
    if (step < nsteps) mfma16(a0, b0);
    __syncthreads();
  }

  if (DIAG(8)) {
    if (acc[0][0][0] == 123.456f) a.out[0] = 1.f;
    return;
  }
  // ---- epilogue: bias + activation, store --------------------------------
  // C/D map of 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5):
  // a lane owns ONE channel of 16 pixels.  NHWC wants the opposite (one pixel,
  // consecutive channels per lane), so each wave transposes its 32x32 tile
  // through a private LDS patch and stores 16 bytes per lane: 4 store
  // instructions per tile instead of 16.
  float* epi = lds + wave * (32 * EPI_STRIDE);
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    if (!nvalid[j]) continue;
    const int n = ntile[j] * 32 + l31;
    {
      const bool nok = n < a.Cout;
      const float bias = nok ? a.bias[n] : 0.f;
      float beta = 1.f, gamma = 0.f;
      if (nok && (a.act == DSIC_ACT_GDN || a.act == DSIC_ACT_IGDN)) {
        beta = a.beta[n];
        gamma = a.gamma[n];
      }
      const bool wide = (a.Cout & 3) == 0;
#pragma unroll
      for (int m = 0; m < MTW; ++m) {
        const int mt = NARROW ? wave : m;
        if (wide) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int rr = (e & 3) + 8 * (e >> 2) + 4 * h;
            epi[rr * EPI_STRIDE + l31] = apply_act(__fadd_rn(acc[m][j][e], bias), a.act, beta, gamma);
          }
          // same-wave LDS round trip: program order + lgkmcnt wait suffice
          __builtin_amdgcn_s_waitcnt(0xc07f);
          __builtin_amdgcn_wave_barrier();
          const int c4 = (lane & 7) * 4;
          const int nn = ntile[j] * 32 + c4;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int rr = (lane >> 3) + 8 * i;
            const floatx4 v = *(const floatx4*)(epi + rr * EPI_STRIDE + c4);
            const int r = mt * 32 + rr;
            const int ox = ox0 + r % TOW;
            const int oy = oy0 + (r / TOW) % TOH;
            const int ni = n0 + r / (TOW * TOH);
            if (ni < a.B && oy < a.Ho && ox < a.Wo && nn < a.Cout) {
              const size_t o = (((size_t)ni * a.oH + (oy * a.os + py)) * a.oW + (ox * a.os + px)) * a.Cout + nn;
              *(floatx4*)(a.out + o) = v;
            }
          }
          __builtin_amdgcn_s_waitcnt(0xc07f);
          __builtin_amdgcn_wave_barrier();
        } else if (nok) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int r = mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            const int ox = ox0 + r % TOW;
            const int oy = oy0 + (r / TOW) % TOH;
            const int ni = n0 + r / (TOW * TOH);
            if (ni < a.B && oy < a.Ho && ox < a.Wo) {
              const size_t o = (((size_t)ni * a.oH + (oy * a.os + py)) * a.oW + (ox * a.os + px)) * a.Cout + n;
              a.out[o] = apply_act(__fadd_rn(acc[m][j][e], bias), a.act, beta, gamma);
            }
          }
        }
      }
    }
  }
}

template <int WIN, int S, int CK, int NTW, bool NARROW>
static int launch_conv(ConvArgs& a, int nphase, int colblocks, hipStream_t st) {
  // tile shape by output-grid width: 16x8x1, 8x8x2, 4x4x8 (cols x rows x images)
  constexpr int CKS = (WIN == 5) ? 8 : CK;  // keep the 4x4x8 window under 64 KB of LDS
  dim3 block(256);
  if (a.Wo > 8) {
    a.tiles_x = ceil_div(a.Wo, 16);
    a.tiles_y = ceil_div(a.Ho, 8);
    dim3 grid(a.tiles_x * a.tiles_y * a.B, nphase, colblocks);
    hipLaunchKernelGGL((conv_igemm_kernel<WIN, S, 16, 8, 1, CK, NTW, NARROW>), grid, block, 0, st, a);
  } else if (a.Wo > 4) {
    a.tiles_x = ceil_div(a.Wo, 8);
    a.tiles_y = ceil_div(a.Ho, 8);
    dim3 grid(a.tiles_x * a.tiles_y * ceil_div(a.B, 2), nphase, colblocks);
    hipLaunchKernelGGL((conv_igemm_kernel<WIN, S, 8, 8, 2, CK, NTW, NARROW>), grid, block, 0, st, a);
  } else {
    a.tiles_x = ceil_div(a.Wo, 4);
    a.tiles_y = ceil_div(a.Ho, 4);
    dim3 grid(a.tiles_x * a.tiles_y * ceil_div(a.B, 8), nphase, colblocks);
    hipLaunchKernelGGL((conv_igemm_kernel<WIN, S, 4, 4, 8, CKS, NTW, NARROW>), grid, block, 0, st, a);
  }
  return check_launch("conv_igemm");
}

template <int WIN, int S, int CK>
static int run_conv_ck(ConvArgs& a, int nphase, hipStream_t st) {
  const int ntiles = a.CoutP / 32;
  if (ntiles == 1) return launch_conv<WIN, S, CK, 1, true>(a, nphase, 1, st);
  if (ntiles <= 4) return launch_conv<WIN, S, CK, 1, false>(a, nphase, 1, st);
  if (ntiles <= 8) {
    // Two column tiles per wave halve the A traffic, but a small grid (the 16x16 latent
    // layer: 128 workgroups at B=64) leaves CUs idle: then split the columns over blockIdx.z.
    const int tw = a.Wo > 8 ? 16 : (a.Wo > 4 ? 8 : 4), th = a.Wo > 4 ? 8 : 4, tn = 128 / (tw * th);
    const long wgs = (long)ceil_div(a.Wo, tw) * ceil_div(a.Ho, th) * ceil_div(a.B, tn) * nphase;
    if (wgs < 512) return launch_conv<WIN, S, CK, 1, false>(a, nphase, ceil_div(ntiles, 4), st);
    return launch_conv<WIN, S, CK, 2, false>(a, nphase, 1, st);
  }
  set_error("conv: Cout=%d too wide (max 256)", a.Cout);
  return DSIC_EINVAL;
}

static int run_conv(ConvArgs& a, int win, int stride, int nphase, hipStream_t st) {
#if DSIC_DIAG
  {
    const char* d = getenv("DSIC_DBG");
    a.dbg = d ? atoi(d) : 0;
  }
#endif
  if (win == 3 && stride == 1)
    return a.Cin % 32 == 0 ? run_conv_ck<3, 1, 32>(a, nphase, st) : run_conv_ck<3, 1, 8>(a, nphase, st);
  if (win == 5 && stride == 2)
    return a.Cin % 16 == 0 ? run_conv_ck<5, 2, 16>(a, nphase, st) : run_conv_ck<5, 2, 8>(a, nphase, st);
  set_error("conv: unsupported geometry win=%d stride=%d", win, stride);
  return DSIC_EINVAL;
}

}  // namespace dsic

using namespace dsic;

extern "C" int dsic_conv2d_nhwc(const float* in, const float* w_packed, const float* bias,
                                const float* beta, const float* gamma, float* out, int B, int H,
                                int W, int CinP, int Cout, int k, int stride, int act,
                                void* stream) {
  DSIC_REQUIRE(in && w_packed && bias && out, "conv2d: null pointer");
  DSIC_REQUIRE(B > 0 && H > 0 && W > 0, "conv2d: empty tensor B=%d H=%d W=%d", B, H, W);
  DSIC_REQUIRE(CinP > 0 && CinP % 8 == 0, "conv2d: CinP=%d must be a positive multiple of 8", CinP);
  DSIC_REQUIRE(Cout > 0, "conv2d: Cout=%d", Cout);
  DSIC_REQUIRE((k == 3 && stride == 1) || (k == 5 && stride == 2),
               "conv2d: (k,stride)=(%d,%d) not in {(3,1),(5,2)}", k, stride);
  DSIC_REQUIRE(act >= 0 && act <= 3, "conv2d: act=%d", act);
  DSIC_REQUIRE(!(act == DSIC_ACT_GDN || act == DSIC_ACT_IGDN) || (beta && gamma),
               "conv2d: GDN needs beta and gamma");
  DSIC_REQUIRE((int64_t)8 * H * W * CinP < (int64_t)1 << 31, "conv2d: image too large");
  ConvArgs a{};
  a.in = in; a.w = w_packed; a.bias = bias; a.beta = beta; a.gamma = gamma; a.out = out;
  a.B = B; a.H = H; a.W = W; a.Cin = CinP;
  a.Ho = ceil_div(H, stride); a.Wo = ceil_div(W, stride);
  a.oH = a.Ho; a.oW = a.Wo; a.Cout = Cout; a.CoutP = round_up(Cout, 32);
  a.os = 1; a.transposed = 0; a.act = act;
  return run_conv(a, k, stride, 1, (hipStream_t)stream);
}

extern "C" int dsic_conv_transpose2d_nhwc(const float* in, const float* w_packed,
                                          const float* bias, const float* beta,
                                          const float* gamma, float* out, int B, int H, int W,
                                          int Cin, int Cout, int act, void* stream) {
  DSIC_REQUIRE(in && w_packed && bias && out, "convT: null pointer");
  DSIC_REQUIRE(B > 0 && H > 0 && W > 0, "convT: empty tensor");
  DSIC_REQUIRE(Cin > 0 && Cin % 8 == 0, "convT: Cin=%d must be a positive multiple of 8", Cin);
  DSIC_REQUIRE(Cout > 0 && Cout % 8 == 0, "convT: Cout=%d must be a positive multiple of 8", Cout);
  DSIC_REQUIRE(act >= 0 && act <= 3, "convT: act=%d", act);
  DSIC_REQUIRE(!(act == DSIC_ACT_GDN || act == DSIC_ACT_IGDN) || (beta && gamma),
               "convT: IGDN needs beta and gamma");
  DSIC_REQUIRE((int64_t)8 * H * W * Cin < (int64_t)1 << 31, "convT: image too large");
  ConvArgs a{};
  a.in = in; a.w = w_packed; a.bias = bias; a.beta = beta; a.gamma = gamma; a.out = out;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin;
  a.Ho = H; a.Wo = W; a.oH = 2 * H; a.oW = 2 * W; a.Cout = Cout; a.CoutP = round_up(Cout, 32);
  a.os = 2; a.transposed = 1; a.act = act;
  return run_conv(a, 3, 1, 4, (hipStream_t)stream);
}
