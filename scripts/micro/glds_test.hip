// LDS-DMA facts the ring-buffered pair kernel relies on (gfx950), checked on the device:
//  1. `global_load_lds_dwordx4 v_off, s[base:base+1] offset:imm` with M0 = wave-uniform LDS byte address: lane l's 16 bytes land
//     at M0 + imm + 16 l -- the immediate offset advances the global AND the LDS address -- (also for M0 >= 64 KiB: a
//     4-wavefront workgroup with 20 KiB per wavefront reaches 80 KiB)
//  2. lanes masked out of EXEC write nothing (the bytes behind the active lanes keep their old contents)
//  3. `global_load_lds_dword` (4 bytes per lane) the same
//  4. the issuing wavefront's `s_waitcnt vmcnt(N)` orders its own ds_read behind the DMA, counted in order together with stores
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void glds16(unsigned lds_addr, const void* base, unsigned lane_off) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072" ::"s"(lds_addr), "v"(lane_off), "s"(base) : "memory");
}
__device__ __forceinline__ void glds16_masked(unsigned lds_addr, const void* base, unsigned lane_off) {  // lanes 0..47
  asm volatile("s_mov_b32 m0, %0\n\ts_bfm_b64 exec, 48, 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b64 exec, -1" ::"s"(lds_addr), "v"(lane_off), "s"(base) : "memory");
}
__device__ __forceinline__ void glds4_masked(unsigned lds_addr, const void* base, unsigned lane_off) {  // lanes 0..8
  asm volatile("s_mov_b32 m0, %0\n\ts_bfm_b64 exec, 9, 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b64 exec, -1" ::"s"(lds_addr), "v"(lane_off), "s"(base) : "memory");
}

__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ src, float* __restrict__ out, float* __restrict__ junk) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned wbase = (unsigned)wid * 20480u;
  float* mine = reinterpret_cast<float*>(smem + wbase);
  for (int i = lane; i < 5120; i += 64) mine[i] = -1.f;
  __syncthreads();
  const float* row = src + (size_t)(blockIdx.x * 4 + wid) * 1024;
  // slot at +16384 (wavefront 3: 61440 + 16384 = 77824 > 64 KiB)
  glds16(wbase + 16384u, row, (unsigned)lane * 16u);  // two instructions: offset 0 and offset 3072 (global and LDS)
  glds16_masked(wbase + 16384u + 1024u, row + 256, (unsigned)lane * 16u);
  glds4_masked(wbase + 16384u + 2048u, row + 512, (unsigned)lane * 4u);
  // 3 stores after the DMA: vmcnt(3) must cover the three DMA instructions
  junk[(size_t)blockIdx.x * 256 + threadIdx.x] = 1.f;
  junk[(size_t)(gridDim.x + blockIdx.x) * 256 + threadIdx.x] = 2.f;
  junk[(size_t)(2 * gridDim.x + blockIdx.x) * 256 + threadIdx.x] = 3.f;
  asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  float* o = out + (size_t)(blockIdx.x * 4 + wid) * 1024;
  const float* slot = reinterpret_cast<const float*>(smem + wbase + 16384u);
  for (int i = lane; i < 1024; i += 64) o[i] = slot[i];
}

int main() {
  const int blocks = 2048;
  const size_t n = (size_t)blocks * 4 * 1024;
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (float)(i % 100003);
  float *src, *out, *junk;
  CK(hipMalloc(&src, n * 4 + 64)); CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&junk, (size_t)3 * blocks * 256 * 4));
  CK(hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 81920));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(out, 0, n * 4));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 81920, 0, src, out, junk);
    CK(hipDeviceSynchronize());
    std::vector<float> r(n);
    CK(hipMemcpy(r.data(), out, n * 4, hipMemcpyDeviceToHost));
    size_t bad = 0, first = 0;
    for (size_t w = 0; w < (size_t)blocks * 4; ++w)
      for (int i = 0; i < 1024; ++i) {
        float want;
        if (i < 256) want = h[w * 1024 + i];                          // full instruction
        else if (i < 256 + 192) want = h[w * 1024 + i];               // 48 lanes x 4 floats
        else if (i < 512) want = -1.f;                                // masked lanes: untouched
        else if (i < 512 + 9) want = h[w * 1024 + i];                 // 9 lanes x 1 float
        else if (i >= 768) want = h[w * 1024 + i];                    // the offset:3072 instruction
        else want = -1.f;
        if (r[w * 1024 + i] != want) { if (!bad) first = w * 1024 + i; ++bad; }
      }
    printf("rep %d: %zu mismatches of %zu%s\n", rep, bad, n, bad ? "" : "  (all four facts hold)");
    if (bad) printf("  first at wave %zu float %zu: got %g\n", first / 1024, first % 1024, r[first]);
  }
  return 0;
}
