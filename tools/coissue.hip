// Microbenchmark (diagnostic): can VALU / LDS / transcendental instructions of one wave issue while two
// other waves of the same SIMD keep the MFMA pipe busy with fp32 32x32x2 MFMAs?
// 768 threads: waves 0..7 run dependent MFMA clusters (two waves per SIMD), waves 8..11 run
// `mode`: 0 = nothing, 1 = independent VALU fmas, 2 = LDS reads, 3 = transcendental (rsq).
//   hipcc --offload-arch=gfx950 -O3 [-DHPRIO=3] [-DPACE=n] [-DNMW=4] [-DCHAIN=1] -o tools/coissue tools/coissue.hip
// Measured on MI355X (round 1): the MFMA waves run at exactly 64 cycles per MFMA whatever the third
// wave does, and the third wave makes NO progress until they finish (helper time = MFMA time + its
// own stand-alone time), with or without s_setprio 3 on it: a SIMD issues one VALU-class
// instruction at a time and an MFMA waiting for the matrix pipe holds that port.  Pacing the MFMA
// waves with s_nop (PACE) frees the port but costs more MFMA time than it gives.  Consequence for
// conv_wino.hip: helper-wave VALU work is never free, it is paid in MFMA time 1:1.
// The same holds with ONE MFMA wave per SIMD (-DNMW=4), even when every MFMA depends on the previous one
// (-DCHAIN=1): that wave alone keeps the pipe 100 % busy (64.0 cycles per MFMA) and the VALU wave beside it
// still makes no progress.  Only instructions of the MFMA-issuing wave itself ride in an MFMA's shadow.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#ifndef PACE
#define PACE -1
#endif
#ifndef NMW
#define NMW 8   // MFMA waves per workgroup (8 = two per SIMD, 4 = one per SIMD)
#endif
#ifndef CHAIN
#define CHAIN 0 // 1: every MFMA depends on the previous one (single accumulator)
#endif
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(768) void k(long long* out, float* sink, int n_mfma, int n_valu, int mode, int run_mfma) {
  __shared__ float lds[4096];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  lds[tid] = tid;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
#ifdef HPRIO
  if (wave >= 8) __builtin_amdgcn_s_setprio(HPRIO);
#endif
  if (wave < 8) {
    if (run_mfma && wave < NMW) {
      floatx16 acc[8];
      for (int p = 0; p < 8; ++p) for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;
      float a = lane * 0.001f, b = 1.0f + lane;
      for (int i = 0; i < n_mfma; ++i) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            acc[CHAIN ? 0 : p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[CHAIN ? 0 : p], 0, 0, 0);
            if (PACE >= 0) asm volatile("s_nop 15\n\ts_nop %0" : : "n"(PACE));  // scalar-side pacing: keeps the next MFMA off the VALU port
          }
        }
      }
      float r = 0.f;
      for (int p = 0; p < 8; ++p) for (int e = 0; e < 16; ++e) r += acc[p][e];
      sink[blockIdx.x * 768 + tid] = r;
    }
  } else if (mode == 1) {
    float x[16];
    for (int j = 0; j < 16; ++j) x[j] = lane + j;
    for (int i = 0; i < n_valu; ++i) {
#pragma unroll
      for (int j = 0; j < 16; ++j) x[j] = __builtin_fmaf(x[j], 1.0001f, 0.5f);
    }
    float r = 0.f;
    for (int j = 0; j < 16; ++j) r += x[j];
    sink[blockIdx.x * 768 + tid] = r;
  } else if (mode == 2) {
    float r = 0.f;
    for (int i = 0; i < n_valu; ++i) {
#pragma unroll
      for (int j = 0; j < 16; ++j) r += lds[(tid + j * 64 + i) & 4095];
    }
    sink[blockIdx.x * 768 + tid] = r;
  } else if (mode == 3) {
    float x[16];
    for (int j = 0; j < 16; ++j) x[j] = lane + j + 1.f;
    for (int i = 0; i < n_valu; ++i) {
#pragma unroll
      for (int j = 0; j < 16; ++j) x[j] = __builtin_amdgcn_rsqf(x[j]) + 1.0f;
    }
    float r = 0.f;
    for (int j = 0; j < 16; ++j) r += x[j];
    sink[blockIdx.x * 768 + tid] = r;
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * 12 + wave] = t1 - t0;
}
int main() {
  long long* out; float* sink;
  hipMalloc(&out, 256 * 12 * 8); hipMalloc(&sink, 256 * 768 * 4);
  long long h[256 * 12];
  const int NM = 2000, NV = 20000;
  for (int mode = 0; mode < 4; ++mode)
    for (int run_mfma = 0; run_mfma < 2; ++run_mfma) {
      if (mode == 0 && !run_mfma) continue;
      for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k, dim3(256), dim3(768), 0, 0, out, sink, NM, NV, mode, run_mfma);
        hipDeviceSynchronize();
      }
      hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
      double m = 0, v = 0;
      for (int b = 0; b < 256; ++b) { for (int w = 0; w < NMW; ++w) m += h[b * 12 + w]; for (int w = 8; w < 12; ++w) v += h[b * 12 + w]; }
      m /= 256 * NMW; v /= 256 * 4;
      printf("pace %d mode %d mfma %d: MFMA wave cycles %.0f (ideal %d for 2 waves/SIMD)   helper wave cycles %.0f  (%d x16 ops: %.2f cyc/op)\n",
             PACE, mode, run_mfma, m, NM * 32 * 64 * 2, v, NV, v / (NV * 16.0));
    }
  return 0;
}
