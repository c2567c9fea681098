// Diagnostic: sustained v_mfma_f32_32x32x2_f32 rate with different operand-feeding structures.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

#define MFMA16(A, B)                                                                      \
  _Pragma("unroll") for (int s = 0; s < 4; ++s) _Pragma("unroll") for (int m = 0; m < 4; ++m) \
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[m][s], B[s], acc[m], 0, 0, 0);

// MODE 0: no LDS reads. 1: read then use (exposed). 2: double buffer with movs.
// 3: ping-pong unrolled x2 (no movs). 4: ping-pong + setprio. 5: ping-pong, reads interleaved after 4 MFMAs
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, const float* in) {
  __shared__ float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = in[i & 1023];
  __syncthreads();
  floatx16 acc[4];
  for (int m = 0; m < 4; ++m) for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;
  floatx4 a[4], a2[4], b;
  const int base = threadIdx.x * 4;
  for (int m = 0; m < 4; ++m) a[m] = *(floatx4*)(lds + ((base + m * 1024) & 8191));
  for (int m = 0; m < 4; ++m) a2[m] = a[m];
  b = *(floatx4*)(lds + ((base + 512) & 8191));
  long long t0 = __builtin_amdgcn_s_memtime();
  long long r0 = __builtin_amdgcn_s_memrealtime();
  if (MODE == 0) {
    for (int it = 0; it < iters; ++it) { MFMA16(a, b) }
  } else if (MODE == 1) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < 4; ++m) a[m] = *(floatx4*)(lds + ((base + m * 1024 + it * 36) & 8188));
      MFMA16(a, b)
    }
  } else if (MODE == 2) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < 4; ++m) a2[m] = *(floatx4*)(lds + ((base + m * 1024 + it * 36) & 8188));
      MFMA16(a, b)
#pragma unroll
      for (int m = 0; m < 4; ++m) a[m] = a2[m];
    }
  } else if (MODE == 3 || MODE == 4) {
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
      for (int m = 0; m < 4; ++m) a2[m] = *(floatx4*)(lds + ((base + m * 1024 + it * 36) & 8188));
      if (MODE == 4) __builtin_amdgcn_s_setprio(1);
      MFMA16(a, b)
      if (MODE == 4) __builtin_amdgcn_s_setprio(0);
#pragma unroll
      for (int m = 0; m < 4; ++m) a[m] = *(floatx4*)(lds + ((base + m * 1024 + it * 36 + 36) & 8188));
      if (MODE == 4) __builtin_amdgcn_s_setprio(1);
      MFMA16(a2, b)
      if (MODE == 4) __builtin_amdgcn_s_setprio(0);
    }
  } else if (MODE == 5) {
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        a2[s] = *(floatx4*)(lds + ((base + s * 1024 + it * 36) & 8188));
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][s], b[s], acc[m], 0, 0, 0);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        a[s] = *(floatx4*)(lds + ((base + s * 1024 + it * 36 + 36) & 8188));
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[m][s], b[s], acc[m], 0, 0, 0);
      }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int m = 0; m < 4; ++m) for (int e = 0; e < 16; ++e) s += acc[m][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    out[0] = (float)(t1 - t0);
    out[1] = (float)(r1 - r0);
  }
}

template <int MODE>
void run(float* out, float* in, hipEvent_t e0, hipEvent_t e1) {
  for (int wgs = 1; wgs <= 3; ++wgs) {
    int iters = 12000 / wgs;
    int grid = 256 * wgs;
    float best = 1e9; float h[2];
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, in);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
      hipMemcpy(h, out, 8, hipMemcpyDeviceToHost);
    }
    double flops = (double)grid * 4 * iters * 16 * 4096.0;
    printf("mode=%d wgs/CU=%d: %.1f TFLOP/s  clk=%.0f MHz\n", MODE, wgs, flops / best / 1e9, h[0] / h[1] * 100.0);
  }
}

int main() {
  float *out, *in;
  (void)hipMalloc(&out, 256 * 4096 * 4);
  (void)hipMalloc(&in, 4096);
  float hin[1024];
  for (int i = 0; i < 1024; ++i) hin[i] = (float)((i * 2654435761u) >> 8) / 16777216.f - 0.5f;
  (void)hipMemcpy(in, hin, 4096, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  run<0>(out, in, e0, e1);
  run<1>(out, in, e0, e1);
  run<2>(out, in, e0, e1);
  run<3>(out, in, e0, e1);
  run<4>(out, in, e0, e1);
  run<5>(out, in, e0, e1);
  return 0;
}
