/* A plain C client of libdsic_hip.so: no Python, no torch - raw device pointers from the HIP runtime
 * API, the entry points of include/dsic_hip.h, results checked against loops on the host.
 * Built by __graft_entry__.build() (gcc), run by tests/test_gpu_abi.py on the GPU box. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dsic_hip.h"

#define CHECK_HIP(x)                                                         \
  do {                                                                       \
    hipError_t e_ = (x);                                                     \
    if (e_ != hipSuccess) {                                                  \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
      return 2;                                                              \
    }                                                                        \
  } while (0)
#define CHECK_DSIC(x)                                                        \
  do {                                                                       \
    int r_ = (x);                                                            \
    if (r_ != DSIC_OK) {                                                     \
      fprintf(stderr, "%s -> %d: %s\n", #x, r_, dsic_last_error());          \
      return 3;                                                              \
    }                                                                        \
  } while (0)

static float frand(unsigned* s) {
  *s = *s * 1664525u + 1013904223u;
  return (float)((*s >> 8) & 0xffff) / 65536.0f - 0.5f;
}

int main(void) {
  enum { B = 2, H = 12, W = 20, Cin = 32, Cout = 64 };
  const size_t n_in = (size_t)B * H * W * Cin, n_w = (size_t)Cout * Cin * 9, n_out = (size_t)B * H * W * Cout;
  float *x = malloc(n_in * 4), *w = malloc(n_w * 4), *bias = malloc(Cout * 4), *ref = malloc(n_out * 4);
  float *got = malloc(n_out * 4);
  unsigned seed = 12345u;
  for (size_t i = 0; i < n_in; ++i) x[i] = frand(&seed);
  for (size_t i = 0; i < n_w; ++i) w[i] = 0.1f * frand(&seed);
  for (int i = 0; i < Cout; ++i) bias[i] = frand(&seed);
  /* host reference: conv 3x3 stride 1 pad 1 + bias + ReLU, NHWC activations, OIHW weights (layers.py:29-31) */
  for (int b = 0; b < B; ++b)
    for (int y = 0; y < H; ++y)
      for (int xx = 0; xx < W; ++xx)
        for (int o = 0; o < Cout; ++o) {
          double acc = bias[o];
          for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
              const int iy = y + ky - 1, ix = xx + kx - 1;
              if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
              for (int c = 0; c < Cin; ++c)
                acc += (double)x[(((size_t)b * H + iy) * W + ix) * Cin + c] * w[(((size_t)o * Cin + c) * 3 + ky) * 3 + kx];
            }
          ref[(((size_t)b * H + y) * W + xx) * Cout + o] = acc > 0.0 ? (float)acc : 0.f;
        }

  float *dx, *dw, *db, *dout, *dpack, *dwino;
  void* ticket;
  CHECK_HIP(hipMalloc((void**)&dx, n_in * 4));
  CHECK_HIP(hipMalloc((void**)&dw, n_w * 4));
  CHECK_HIP(hipMalloc((void**)&db, Cout * 4));
  CHECK_HIP(hipMalloc((void**)&dout, n_out * 4));
  CHECK_HIP(hipMalloc((void**)&dpack, (size_t)dsic_packed_conv_weight_floats(Cout, Cin, 3) * 4));
  CHECK_HIP(hipMalloc((void**)&dwino, (size_t)dsic_wino_weight_floats(Cout, Cin) * 4));
  CHECK_HIP(hipMalloc(&ticket, 16));
  CHECK_HIP(hipMemset(ticket, 0, 16));
  CHECK_HIP(hipMemcpy(dx, x, n_in * 4, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(dw, w, n_w * 4, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(db, bias, Cout * 4, hipMemcpyHostToDevice));

  int fail = 0;
  for (int path = 0; path < 2; ++path) {
    CHECK_HIP(hipMemset(dout, 0, n_out * 4));
    if (path == 0) { /* direct implicit GEMM */
      CHECK_DSIC(dsic_pack_conv_weight(dw, dpack, Cout, Cin, 3, NULL));
      CHECK_DSIC(dsic_conv2d_nhwc(dx, dpack, db, NULL, NULL, dout, B, H, W, Cin, Cout, 3, 1, DSIC_ACT_RELU, NULL));
    } else { /* Winograd, twice on the same ticket (the kernel re-arms it) */
      CHECK_DSIC(dsic_pack_wino_weight(dw, dwino, Cout, Cin, NULL));
      for (int rep = 0; rep < 2; ++rep)
        CHECK_DSIC(dsic_conv3x3_wino_nhwc(dx, dwino, db, NULL, NULL, dout, B, H, W, Cin, Cout, DSIC_ACT_RELU, 0, 0,
                                          ticket, NULL));
    }
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(got, dout, n_out * 4, hipMemcpyDeviceToHost));
    double worst = 0.0;
    for (size_t i = 0; i < n_out; ++i) {
      const double d = fabs((double)got[i] - ref[i]);
      if (d > worst) worst = d;
    }
    printf("path %d: max |diff| = %.3g\n", path, worst);
    if (!(worst < 2e-5)) fail = 1;
  }
  /* ---- the default arithmetic: transformed weights split into bf16 planes, split-bf16 Winograd kernel; a layer
   * large enough for the 64-tile two-pass kernel (64x64, Cin = Cout = 64: 16 work items per image) ------------- */
  {
    enum { B2 = 1, H2 = 64, W2 = 64, C2 = 64 };
    const size_t n2 = (size_t)B2 * H2 * W2 * C2, nw2 = (size_t)C2 * C2 * 9;
    float *x2 = malloc(n2 * 4), *w2 = malloc(nw2 * 4), *ref2 = malloc(n2 * 4), *got2 = malloc(n2 * 4);
    for (size_t i = 0; i < n2; ++i) x2[i] = frand(&seed);
    for (size_t i = 0; i < nw2; ++i) w2[i] = 0.1f * frand(&seed);
    double refmax = 0.0;
    for (int y = 0; y < H2; ++y)
      for (int xx = 0; xx < W2; ++xx)
        for (int o = 0; o < C2; ++o) {
          double acc = bias[o];
          for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
              const int iy = y + ky - 1, ix = xx + kx - 1;
              if (iy < 0 || iy >= H2 || ix < 0 || ix >= W2) continue;
              for (int c = 0; c < C2; ++c)
                acc += (double)x2[((size_t)iy * W2 + ix) * C2 + c] * w2[(((size_t)o * C2 + c) * 3 + ky) * 3 + kx];
            }
          ref2[((size_t)y * W2 + xx) * C2 + o] = (float)acc;
          if (fabs(acc) > refmax) refmax = fabs(acc);
        }
    float *dx2, *dw2, *dout2, *du32;
    void* dplanes;
    CHECK_HIP(hipMalloc((void**)&dx2, n2 * 4));
    CHECK_HIP(hipMalloc((void**)&dw2, nw2 * 4));
    CHECK_HIP(hipMalloc((void**)&dout2, n2 * 4));
    CHECK_HIP(hipMalloc((void**)&du32, (size_t)dsic_wino_weight_floats(C2, C2) * 4));
    CHECK_HIP(hipMalloc(&dplanes, (size_t)dsic_wino_bf16_weight_bytes(C2, C2)));
    CHECK_HIP(hipMemcpy(dx2, x2, n2 * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(dw2, w2, nw2 * 4, hipMemcpyHostToDevice));
    CHECK_DSIC(dsic_pack_wino_weight(dw2, du32, C2, C2, NULL));
    CHECK_DSIC(dsic_split_wino_weight_bf16(du32, dplanes, C2, C2, 1, NULL));
    CHECK_DSIC(dsic_conv3x3_wino_bf16_nhwc(dx2, dplanes, db, NULL, NULL, dout2, B2, H2, W2, C2, C2, DSIC_ACT_NONE, 0, 0, 0,
                                           0, ticket, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(got2, dout2, n2 * 4, hipMemcpyDeviceToHost));
    double worst = 0.0;
    for (size_t i = 0; i < n2; ++i) {
      const double d = fabs((double)got2[i] - ref2[i]);
      if (d > worst) worst = d;
    }
    printf("split-bf16 Winograd (two-pass kernel: %d): max |diff| = %.3g of |ref|max %.3g\n",
           dsic_wino_bf16_m64(H2, W2, C2, 1), worst, refmax);
    if (!(worst < 4 * 2e-5 * (refmax > 1.0 ? refmax : 1.0))) fail = 1; /* 4x the fp32 class: dropped 2^-16 cross terms */
  }
  /* ---- entropy path: Gaussian tables -> range encoder -> range decoder on a small integer latent
   * (eval_selfcontained_entropy.py:36-48, 88-96) -------------------------------------------------------------- */
  {
    enum { EB = 2, EN = 4, EHW = 48, EM = 4, EHWY = 48, LMAX = 64, TAIL = 10 };
    const int nz = EN * EHW;
    float *z = malloc((size_t)EB * nz * 4), *zback = malloc((size_t)EB * nz * 4);
    for (int i = 0; i < EB * nz; ++i) z[i] = (float)((int)(frand(&seed) * 14.0f));   /* integers in [-7, 7] */
    const float sigma_z[EN] = {0.7f, 1.5f, 2.5f, 4.0f};
    const int64_t cap = 4 * ((2 * nz + 64 + 3) / 4);   /* worst case 2 bytes per symbol */
    float *dz, *dsig, *dzback;
    int *dmeta, *derr, *dlen;
    uint16_t* dtab;
    uint8_t* dbytes;
    int meta[4 * EB], err = 0, len[2 * EB];
    CHECK_HIP(hipMalloc((void**)&dz, (size_t)EB * nz * 4));
    CHECK_HIP(hipMalloc((void**)&dzback, (size_t)EB * nz * 4));
    CHECK_HIP(hipMalloc((void**)&dsig, EN * 4));
    CHECK_HIP(hipMalloc((void**)&dmeta, sizeof meta));
    CHECK_HIP(hipMalloc((void**)&derr, 4));
    CHECK_HIP(hipMalloc((void**)&dlen, sizeof len));
    CHECK_HIP(hipMalloc((void**)&dtab, (size_t)EB * EN * LMAX * 2));
    CHECK_HIP(hipMalloc((void**)&dbytes, (size_t)EB * 2 * cap));
    CHECK_HIP(hipMemcpy(dz, z, (size_t)EB * nz * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(dsig, sigma_z, sizeof sigma_z, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemset(derr, 0, 4));
    CHECK_HIP(hipMemset(dbytes, 0, (size_t)EB * 2 * cap));
    CHECK_HIP(hipMemset(dlen, 0, sizeof len));
    /* the same latent serves as "y" and "z" of the call (both strings use the Gaussian tables here) */
    CHECK_DSIC(dsic_latent_support(dz, dz, dmeta, EB, nz, nz, TAIL, NULL));
    CHECK_DSIC(dsic_cdf_tables_gauss(dsig, dmeta, dtab, EB, EN, LMAX, derr, NULL));
    CHECK_DSIC(dsic_range_encode(dz, dz, dmeta, dtab, dtab, LMAX, EB, EM, EHWY, EN, EHW, dbytes, cap, cap, dlen, derr, 1, 0,
                                 NULL));
    /* decode the z string (offset 0 of every image's [z | y] buffer) */
    CHECK_DSIC(dsic_range_decode(dbytes, 2 * cap, dlen, 2, 0, dmeta, 2, dtab, LMAX, EB, EN, EHW, 0, dzback, derr, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(zback, dzback, (size_t)EB * nz * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(&err, derr, 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(len, dlen, sizeof len, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(meta, dmeta, sizeof meta, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < EB * nz; ++i) bad += zback[i] != z[i];
    printf("range coder: %d + %d bytes for 2 x %d symbols (support %d wide), err %d, %d symbols differ after decode\n",
           len[0], len[2], nz, meta[3], err, bad);
    if (bad || err || len[0] <= 0 || len[0] != len[1] || len[0] > nz) fail = 1;   /* y string = z string: same bytes */
  }
  /* error behaviour: a bad argument is refused with DSIC_EINVAL and a message, nothing is launched */
  const int rc = dsic_conv2d_nhwc(dx, dpack, db, NULL, NULL, dout, B, H, W, Cin, Cout, 3, 2, DSIC_ACT_NONE, NULL);
  if (rc != DSIC_EINVAL || strlen(dsic_last_error()) == 0) {
    fprintf(stderr, "bad (k,stride) was not refused: rc %d\n", rc);
    fail = 1;
  }
  if (dsic_conv2d_nhwc(dx, dpack, db, NULL, NULL, dout, B, H, W, Cin, Cout, 3, 1, DSIC_ACT_GDN, NULL) != DSIC_EINVAL) {
    fprintf(stderr, "GDN without beta/gamma was not refused\n");
    fail = 1;
  }
  puts(fail ? "c_client FAILED" : "c_client ok");
  return fail;
}
