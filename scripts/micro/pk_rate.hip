// How fast are plain and packed fp32 FMAs on this device, per SIMD, at 1 / 2 / 4 / 8 wavefronts per SIMD?
//   hipcc --offload-arch=gfx950 -O3 pk_rate.hip -o pk_rate.out
// Each lane runs NACC independent accumulator chains of v_fma_f32 (or v_pk_fma_f32 on float2 values) for ITER rounds.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ITER = 4096;

template <int NACC>
__global__ __launch_bounds__(256) void k_fma(float* out, float a, float b) {
  float acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-3f + i;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fmaf(acc[i], a, b);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i];
  if (s == 12345.678f) out[0] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k_pk(float* out, float a, float b) {
  f2 acc[NACC];
  const f2 a2 = {a, a * 1.0001f}, b2 = {b, b * 0.999f};
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f2{threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f + i};
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_elementwise_fma(acc[i], a2, b2);
  }
  f2 s = {0, 0};
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i];
  if (s.x + s.y == 12345.678f) out[0] = s.x;
}
// packed with a wave-uniform (scalar register) multiplier broadcast to both halves, as the CG constants would be
template <int NACC>
__global__ __launch_bounds__(256) void k_pk_s(float* out, const float* __restrict__ c) {
  f2 acc[NACC];
  const float c0 = c[0], c1 = c[1];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f2{threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f + i};
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = acc[i] * c0 + f2{c1, c1};
  }
  f2 s = {0, 0};
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i];
  if (s.x + s.y == 12345.678f) out[0] = s.x;
}
template <typename F>
static double run(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0); for (int r = 0; r < 5; ++r) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}
int main() {
  float* out; hipMalloc(&out, 64);
  float h[2] = {0.999f, 1e-3f}; float* c; hipMalloc(&c, 8); hipMemcpy(c, h, 8, hipMemcpyHostToDevice);
  int ncu = 256; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  for (int wps : {1, 2, 4, 8}) {  // wavefronts per SIMD: blocks of 4 wavefronts, `wps` blocks per CU
    const unsigned grid = ncu * wps;
    const double lanes = (double)grid * 256;
    auto rep = [&](const char* name, double ms, int nacc, int width) {
      const double fma = lanes * nacc * ITER * width;
      printf("wps %d %-22s nacc %2d: %7.3f ms  %7.1f TFLOP/s  (%.2f cycles per wave instruction at 2.4 GHz)\n", wps, name, nacc, ms,
             2 * fma / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)nacc * ITER * wps));
    };
    rep("v_fma_f32", run([&] { hipLaunchKernelGGL(k_fma<8>, dim3(grid), dim3(256), 0, 0, out, 0.999f, 1e-3f); }), 8, 1);
    rep("v_fma_f32", run([&] { hipLaunchKernelGGL(k_fma<2>, dim3(grid), dim3(256), 0, 0, out, 0.999f, 1e-3f); }), 2, 1);
    rep("v_fma_f32", run([&] { hipLaunchKernelGGL(k_fma<1>, dim3(grid), dim3(256), 0, 0, out, 0.999f, 1e-3f); }), 1, 1);
    rep("v_pk_fma_f32", run([&] { hipLaunchKernelGGL(k_pk<8>, dim3(grid), dim3(256), 0, 0, out, 0.999f, 1e-3f); }), 8, 2);
    rep("v_pk_fma_f32", run([&] { hipLaunchKernelGGL(k_pk<2>, dim3(grid), dim3(256), 0, 0, out, 0.999f, 1e-3f); }), 2, 2);
    rep("v_pk_fma_f32", run([&] { hipLaunchKernelGGL(k_pk<1>, dim3(grid), dim3(256), 0, 0, out, 0.999f, 1e-3f); }), 1, 2);
    rep("v_pk_fma_f32 sgpr", run([&] { hipLaunchKernelGGL(k_pk_s<8>, dim3(grid), dim3(256), 0, 0, out, c); }), 8, 2);
  }
  return 0;
}
