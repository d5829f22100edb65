// Pure-MFMA ceilings on this device (no memory traffic): v_mfma_f32_32x32x16_bf16 and v_mfma_f32_32x32x2_f32,
// 1..4 wavefronts per SIMD, independent accumulator chains.  Build + run: hipcc --offload-arch=gfx950 -O3 mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
template <int NACC>
__global__ __launch_bounds__(256) void k_bf16(float* out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x16){0};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 3); b[i] = (__bf16)1.0f; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0];
  if (s == 12345.f) out[0] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k_f32(float* out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x16){0};
  float a = threadIdx.x & 3, b = 1.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0];
  if (s == 12345.f) out[0] = s;
}
template <typename K>
static double run(K kern, int blocks, int iters, double flop_per_mfma, int nacc) {
  float* d; hipMalloc(&d, 4);
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters);
  hipDeviceSynchronize();
  hipEventRecord(s);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  hipFree(d);
  return (double)blocks * 4 * iters * nacc * flop_per_mfma / (ms * 1e-3) / 1e12;
}
int main() {
  const int iters = 4000;
  for (int wps : {1, 2, 4}) {
    const int blocks = 256 * wps;
    printf("bf16 32x32x16: %d wave/SIMD  NACC=2 %.0f TF  NACC=4 %.0f TF\n", wps,
           run(k_bf16<2>, blocks, iters, 2.0 * 32 * 32 * 16, 2), run(k_bf16<4>, blocks, iters, 2.0 * 32 * 32 * 16, 4));
    printf("f32  32x32x2 : %d wave/SIMD  NACC=2 %.0f TF  NACC=4 %.0f TF\n", wps,
           run(k_f32<2>, blocks, iters, 2.0 * 32 * 32 * 2, 2), run(k_f32<4>, blocks, iters, 2.0 * 32 * 32 * 2, 4));
  }
  return 0;
}
