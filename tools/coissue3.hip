// Microbenchmark (diagnostic): do VALU instructions of a helper wave run in the shadow of the bf16 MFMAs of the two
// MFMA waves it shares a SIMD with?  (tools/coissue2.hip answered "no" for the fp32-input MFMA.)
//   768 threads = 12 waves, three per SIMD as in conv_wino_bf16m: waves 0..7 issue v_mfma_f32_32x32x16_bf16 on 8
//   rotating accumulators (NM each), waves 8..11 issue NV independent VALU instructions (kind: 0 v_fma_f32,
//   1 the split sequence cvt_pk / and / sub / cvt_pk).
//   run 0: MFMA waves only      -> cycles per MFMA of a SIMD (32 = pipe saturated)
//   run 1: helper waves only    -> cycles per VALU instruction
//   run 2: both                 -> if the helpers' instructions hide behind the MFMAs, the MFMA waves take as long as in
//                                  run 0 and the helpers finish inside that time; if they are additive, run 2 = run 0 + run 1
//   run 3: MFMA waves only, each step 3 MFMAs followed by KS VALU instructions of the same wave
//   run 4: as run 2 with ONE MFMA wave per SIMD (waves 4..7 idle)     run 5 / 6: run 2 / run 4 with the helpers at s_setprio 3
// Output per run: cycles (s_memtime) until the last MFMA wave / the last helper wave of a workgroup has finished (mean over
// workgroups), the end time of every wave of workgroup 0, the in-kernel clock (s_memtime / s_memrealtime x 100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short shortx8 __attribute__((ext_vector_type(8)));

template <int KIND, int KS>
__global__ __launch_bounds__(768) void k(long long* out, float* sink, int nm, int nv, int run) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  long long t0 = 0, t1 = 0;
  __syncthreads();
  const long long tb = __builtin_amdgcn_s_memtime(), rb = __builtin_amdgcn_s_memrealtime();
  if (wave < 8) {
    floatx16 acc[8];
    for (int p = 0; p < 8; ++p) for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)(1.0f + e); }
    float x[8];
    for (int j = 0; j < 8; ++j) x[j] = lane + j;
    t0 = __builtin_amdgcn_s_memtime();
    if (run != 1 && !(wave >= 4 && (run == 4 || run == 6))) {
      for (int it = 0; it < nm / 24; ++it) {
#pragma unroll
        for (int st = 0; st < 8; ++st) {
#pragma unroll
          for (int s = 0; s < 3; ++s) {
            const int p = (st * 3 + s) & 7;
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[p], 0, 0, 0);
          }
          if (run == 3) {
#pragma unroll
            for (int j = 0; j < KS; ++j) x[j & 7] = __builtin_fmaf(x[j & 7], 1.0001f, 0.5f);
          }
        }
      }
    }
    t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int p = 0; p < 8; ++p) for (int e = 0; e < 16; ++e) s += acc[p][e];
    for (int j = 0; j < 8; ++j) s += x[j];
    if (s == 12345.678f) sink[tid] = s;
  } else {
    float x[16];
    for (int j = 0; j < 16; ++j) x[j] = lane * 0.37f + j;
    unsigned pk[8];
    for (int j = 0; j < 8; ++j) pk[j] = 0;
    t0 = __builtin_amdgcn_s_memtime();
    if (run == 5 || run == 6) __builtin_amdgcn_s_setprio(3);
    if (run == 1 || run == 2 || run >= 4) {
      if (KIND == 0) {
        for (int it = 0; it < nv / 64; ++it) {
#pragma unroll
          for (int j = 0; j < 64; ++j) x[j & 15] = __builtin_fmaf(x[j & 15], 1.0001f, 0.5f);
        }
      } else {
        // 8 instructions per pair of values: cvt_pk hi, shift, and, 2 sub, cvt_pk mid, xor, add
        for (int it = 0; it < nv / 64; ++it) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float v0 = x[2 * j], v1 = x[2 * j + 1];
            unsigned hi;
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(v0), "v"(v1));
            const float h0 = __builtin_bit_cast(float, hi << 16), h1 = __builtin_bit_cast(float, hi & 0xffff0000u);
            const float r0 = v0 - h0, r1 = v1 - h1;
            unsigned mid;
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(mid) : "v"(r0), "v"(r1));
            pk[j] ^= mid;
            x[2 * j] = r0 + v1;     // keeps the chain alive (7th and 8th instruction of the group)
          }
        }
      }
    }
    t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < 16; ++j) s += x[j];
    for (int j = 0; j < 8; ++j) s += (float)pk[j];
    if (s == 12345.678f) sink[tid] = s;
  }
  if (lane == 0) { out[(blockIdx.x * 12 + wave) * 2] = t0 - tb; out[(blockIdx.x * 12 + wave) * 2 + 1] = t1 - tb; }
  __syncthreads();
  if (tid == 0) { out[gridDim.x * 24 + blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - tb; out[gridDim.x * 24 + blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - rb; }
}

template <int KIND, int KS>
static void go(const char* name, int nm, int nv) {
  long long* d; float* sink;
  const int nb = 256;
  hipMalloc(&d, nb * 26 * sizeof(long long));
  hipMalloc(&sink, 768 * sizeof(float));
  long long* h = (long long*)malloc(nb * 26 * sizeof(long long));
  for (int run = 0; run < 7; ++run) {
    if (run == 3 && KS == 0) continue;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL((k<KIND, KS>), dim3(nb), dim3(768), 0, 0, d, sink, nm, nv, run);
      hipEventRecord(e1, 0);
      hipDeviceSynchronize();
      hipEventElapsedTime(&ms, e0, e1);
    }
    hipMemcpy(h, d, nb * 26 * sizeof(long long), hipMemcpyDeviceToHost);
    double m = 0, v = 0, tot = 0, real = 0;
    for (int b = 0; b < nb; ++b) {
      double mm = 0, vv = 0;
      for (int w = 0; w < 8; ++w) mm = h[(b * 12 + w) * 2 + 1] > mm ? h[(b * 12 + w) * 2 + 1] : mm;   // last MFMA wave to finish
      for (int w = 8; w < 12; ++w) vv = h[(b * 12 + w) * 2 + 1] > vv ? h[(b * 12 + w) * 2 + 1] : vv;
      m += mm; v += vv; tot += h[nb * 24 + b * 2]; real += h[nb * 24 + b * 2 + 1];
    }
    m /= nb; v /= nb;
    printf("   block 0, end of each wave (k cycles):");
    for (int w = 0; w < 12; ++w) printf(" %.0f", h[w * 2 + 1] * 1e-3);
    printf("   clock %.2f GHz\n", tot / real * 0.1);
    // two MFMA waves per SIMD: SIMD cycles per MFMA = elapsed / (2 nm)
    printf("%s run %d: MFMA wave %.0f cycles (%.1f per MFMA of the SIMD)   helper wave %.0f cycles (%.2f per VALU instruction)   kernel %.1f us = %.2f ticks/ns\n",
           name, run, m, m / (2.0 * nm), v, v / nv, ms * 1e3, (m > v ? m : v) / (ms * 1e6));
  }
  hipFree(d); hipFree(sink); free(h);
}

int main() {
  const int nm = 24 * 256, nv = 64 * 48 * 16;   // 6144 MFMAs per wave (393 k cycles per SIMD), 49152 VALU instructions
  go<0, 0>("fma      ", nm, nv);
  go<1, 0>("split    ", nm, nv);
  go<0, 12>("same-wave 12 VALU per 3 MFMA", nm, nv);
  return 0;
}
