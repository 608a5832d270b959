// Micro-benchmark (not product code): do FP64 MFMA (v_mfma_f64_16x16x4_f64) and FP64 VALU (v_fma_f64) overlap on gfx950?
//   hipcc --offload-arch=gfx950 -O3 -o f64_overlap f64_overlap.hip && ./f64_overlap
// mode 0: MFMA only   1: VALU only   2: both interleaved in ONE wave   3: 8 waves per CU, waves 0-3 MFMA, 4-7 VALU (2 per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void k(double* out, int iters, double seed) {
  const int wave = threadIdx.x >> 6;
  f64x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (f64x4){seed, seed, seed, seed};
  double a = seed + threadIdx.x, b = seed * 0.5;
  double v[16];
  for (int i = 0; i < 16; ++i) v[i] = seed + i;
  const bool do_mfma = (MODE == 0) || (MODE == 2) || (MODE == 3 && wave < 4);
  const bool do_valu = (MODE == 1) || (MODE == 2) || (MODE == 3 && wave >= 4);
  for (int it = 0; it < iters; ++it) {
    if (do_mfma && do_valu) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[r], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_fma(v[i], 1.0000001, 0.5);   // 16 independent FMAs per MFMA
      }
    } else if (do_mfma) {
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[r], 0, 0, 0);
    } else if (do_valu) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_fma(v[i], 1.0000001, 0.5);
    }
  }
  double s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
float run(int threads, int iters) {
  double* out;
  hipMalloc(&out, 256 * 512 * sizeof(double) * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, 10, 1.0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters, 1.0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipFree(out);
  return ms;
}

int main() {
  const int iters = 200000;
  // one wave per SIMD (256 threads), 4 MFMA (+ 64 FMA) per iteration
  float m0 = run<0>(256, iters), m1 = run<1>(256, iters), m2 = run<2>(256, iters), m3 = run<3>(512, iters);
  const double mf = 4.0 * iters, vf = 64.0 * iters;
  printf("mfma only : %.3f ms  -> %.1f cycles/MFMA at 2.4 GHz\n", m0, m0 * 1e-3 * 2.4e9 / mf);
  printf("valu only : %.3f ms  -> %.2f cycles/v_fma_f64 at 2.4 GHz\n", m1, m1 * 1e-3 * 2.4e9 / vf);
  printf("one wave interleaved (4 MFMA + 64 FMA per iter): %.3f ms (sum of the two alone %.3f, max %.3f)\n", m2, m0 + m1, m0 > m1 ? m0 : m1);
  printf("two waves per SIMD, one MFMA one VALU          : %.3f ms (sum %.3f, max %.3f)\n", m3, m0 + m1, m0 > m1 ? m0 : m1);
  return 0;
}
