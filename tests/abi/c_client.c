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
