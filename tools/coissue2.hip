// Microbenchmark (diagnostic): does VALU work of the MFMA-issuing wave itself ride in the shadow of its
// MFMAs, and does that survive a second wave on the SIMD when the two take turns (barrier hand-over)?
//   512 threads = 8 waves (two per SIMD: w and w+4).  Every step = 4 dependent fp32 32x32x2 MFMAs followed
//   by K independent VALU fmas (K = 0, 8, 16, 32, 48).  A block = 32 steps.
//   mode 0: only waves 0..3 run (one wave per SIMD)
//   mode 1: all 8 waves run the same stream concurrently (interleaved on the SIMD)
//   mode 2: time-sliced: waves 0..3 run a block, barrier, waves 4..7 run a block, barrier, ...
// Output: cycles per MFMA of the SIMD (64 = pipe saturated).
// Measured on MI355X (round 1): strictly additive in every mode - 64 + 4.3 cycles per VALU instruction
// (K = 8: 72.6, 16: 78.9, 32: 91.7, 48: 104.5), with one wave, two interleaved waves or two waves taking
// turns.  fp32-input MFMA and fp32 VALU run at the same FLOP rate (256 flop/clk/CU) and evidently on the same
// lanes: no VALU instruction hides behind a v_mfma_f32_32x32x2_f32, not even one of the same wave.  The bound
// of an fp32 MFMA kernel is therefore MFMA time PLUS the VALU time of every wave on the SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
template <int K>
__global__ __launch_bounds__(512) void k(long long* out, float* sink, int nblocks, int mode) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  floatx16 acc[8];
  for (int p = 0; p < 8; ++p) for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;
  float x[16];
  for (int j = 0; j < 16; ++j) x[j] = lane + j;
  float a = lane * 0.001f, b = 1.0f + lane;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int blk = 0; blk < nblocks; ++blk) {
    for (int half = 0; half < 2; ++half) {
      const bool active = mode == 0 ? grp == 0 : (mode == 1 ? true : grp == half);
      if (active && (mode == 2 || half == 0)) {
#pragma unroll
        for (int st = 0; st < 32; ++st) {
#pragma unroll
          for (int s = 0; s < 4; ++s) acc[st & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[st & 7], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < K; ++j) x[j & 15] = __builtin_fmaf(x[j & 15], 1.0001f, 0.5f);
        }
      }
      if (mode == 2) __syncthreads();
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0.f;
  for (int p = 0; p < 8; ++p) for (int e = 0; e < 16; ++e) r += acc[p][e];
  for (int j = 0; j < 16; ++j) r += x[j];
  sink[blockIdx.x * 512 + tid] = r;
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int K>
void run(long long* out, float* sink) {
  long long h[256 * 8];
  const int NB = 200;
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(k<K>, dim3(256), dim3(512), 0, 0, out, sink, NB, mode);
      (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    double t = 0;
    for (int b = 0; b < 256; ++b) { long long m = 0; for (int w = 0; w < 8; ++w) if (h[b * 8 + w] > m) m = h[b * 8 + w]; t += m; }
    t /= 256;
    const double mfma_per_simd = (mode == 0 ? 1.0 : 2.0) * NB * 32 * 4;
    printf("K=%2d mode %d: %.1f cycles per MFMA on the SIMD (64 = saturated), %.2f VALU per MFMA\n", K, mode,
           t / mfma_per_simd, K / 4.0);
  }
}
int main() {
  long long* out; float* sink;
  (void)hipMalloc(&out, 256 * 8 * 8); (void)hipMalloc(&sink, 256 * 512 * 4);
  run<0>(out, sink); run<8>(out, sink); run<16>(out, sink); run<32>(out, sink); run<48>(out, sink);
  return 0;
}
