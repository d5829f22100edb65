// Prototype (round 6): an o3.Linear evaluated per TILE OF WHOLE ATOM ROWS -- what would the node side cost if the input rows were
// read as contiguous 9 KB runs (LDS-DMA into LDS, once) instead of 128 d-byte pieces per 32-channel stage?
// Shape: cfg-3 linear_2 (2240 -> 576): l=0: 192 -> 64, l=1: 256 -> 64 (x3 components), l=2: 256 -> 64 (x5).  ONE fp16 product per
// block (the real kernels do three on split planes: the matrix pipe is not what this measures), no scaling machinery.
//   hipcc --offload-arch=gfx950 -O3 node_rows_proto.hip -o node_rows_proto.out ; ./node_rows_proto.out [N]
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int DIN = 2240, DOUT = 576, T = 6, ROW = DIN + 4;  // floats per LDS row (pad: rows start 4 banks apart)
// blocks: (d, K, x offset, out offset)
__constant__ int kBlk[3][4] = {{1, 192, 0, 0}, {3, 256, 192, 64}, {5, 256, 192 + 768, 64 + 192}};

__device__ __forceinline__ void glds16(unsigned lds, const void* base, unsigned off, int nl) {
  if (nl >= 64) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(off), "s"(base) : "memory");
  else { unsigned long long keep; asm volatile("s_mov_b64 %0, exec\n\ts_mov_b32 m0, %1\n\ts_bfm_b64 exec, 48, 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, %0" : "=&s"(keep) : "s"(lds), "v"(off), "s"(base) : "memory"); }
}

// wf: per block, per k16, per 32-channel tile: 64 lanes x 8 halfs (A fragment of v_mfma_f32_32x32x16_f16: row = lane % 32 = output
// channel of the tile, k = 8 (lane / 32) + e)
__global__ __launch_bounds__(256, 2) void rows_kernel(const float* __restrict__ x, const h8* __restrict__ wf, float* __restrict__ out, int N) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int a0 = blockIdx.x * T;
  // ---- whole rows of T atoms -> LDS (9 copies of <= 1 KB per row; wave w takes rows w, w + 4)
  for (int r = wv; r < T; r += 4) {
    const int at = a0 + r < N ? a0 + r : N - 1;
    const float* row = x + (size_t)at * DIN;
    for (int c = 0; c < 9; ++c) glds16((unsigned)(r * ROW * 4 + c * 1024), row + c * 256, (unsigned)lane * 16u, c < 8 ? 64 : 48);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (wv >= 3) return;
  const float* xs = reinterpret_cast<const float*>(smem);
  const int d = kBlk[wv][0], K = kBlk[wv][1], xoff = kBlk[wv][2], ooff = kBlk[wv][3];
  const int col = lane & 31, half = lane >> 5;
  const int z = col / d, m = col - z * d;
  const bool cok = z < T;
  const float* xb = xs + (cok ? z : 0) * ROW + xoff + m;
  size_t fbase = 0;
  for (int b = 0; b < wv; ++b) fbase += (size_t)(kBlk[b][1] / 16) * 2 * 64;
  const h8* wfb = wf + fbase + lane;
  f16v acc0 = {0}, acc1 = {0};
  h8 A0 = wfb[0], A1 = wfb[64];
  for (int k16 = 0; k16 < K / 16; ++k16) {
    h8 B;
#pragma unroll
    for (int e = 0; e < 8; ++e) B[e] = (_Float16)(cok ? xb[(k16 * 16 + 8 * half + e) * d] : 0.f);
    const h8 a0f = A0, a1f = A1;
    if (k16 + 1 < K / 16) { A0 = wfb[(size_t)(k16 + 1) * 128]; A1 = wfb[(size_t)(k16 + 1) * 128 + 64]; }
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0f, B, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1f, B, acc1, 0, 0, 0);
  }
  if (cok && a0 + z < N) {
    float* ob = out + (size_t)(a0 + z) * DOUT + ooff + m;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ch = (r & 3) + 8 * (r >> 2) + 4 * half;
      ob[ch * d] = acc0[r];
      ob[(ch + 32) * d] = acc1[r];
    }
  }
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 10125;
  std::vector<float> hx((size_t)N * DIN);
  for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
  const size_t nfrag = (size_t)(192 / 16 + 256 / 16 + 256 / 16) * 2 * 64;
  std::vector<_Float16> hw(nfrag * 8);
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = (_Float16)((float)((i * 40503u) % 200) / 2000.f - 0.05f);
  float *x, *out; h8* wf;
  CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&out, (size_t)N * DOUT * 4)); CK(hipMalloc(&wf, hw.size() * 2));
  CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(wf, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  const int lds = T * ROW * 4;
  CK(hipFuncSetAttribute((const void*)rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const int blocks = (N + T - 1) / T;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(rows_kernel, dim3(blocks), dim3(256), lds, 0, x, wf, out, N);
  CK(hipEventRecord(e0, 0));
  const int reps = 20;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(rows_kernel, dim3(blocks), dim3(256), lds, 0, x, wf, out, N);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  // spot check of one output against the host
  std::vector<float> ho((size_t)N * DOUT);
  CK(hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost));
  double maxerr = 0;
  for (int at : {0, 7, N - 1}) for (int ch : {0, 33, 63}) {
    // block 1 (d = 3), component 2
    double ref = 0;
    for (int k = 0; k < 256; ++k) {
      const int k16 = k / 16, kk = k % 16, hf = kk / 8, e = kk % 8, tile = ch / 32, row = ch % 32;
      const size_t f = ((size_t)(192 / 16) * 2 * 64 + ((size_t)k16 * 2 + tile) * 64 + (hf * 32 + row)) * 8 + e;
      ref += (double)(float)hw[f] * (double)(float)(_Float16)hx[(size_t)at * DIN + 192 + k * 3 + 2];
    }
    const double got = ho[(size_t)at * DOUT + 64 + ch * 3 + 2];
    maxerr = fmax(maxerr, fabs(got - ref));
  }
  printf("N=%d tiles of %d atoms, %d B LDS per workgroup: %.1f us  (%.2f TB/s of input rows)  spot-check max err %.2e\n", N, T, lds, ms * 1e3,
         (double)N * DIN * 4 / (ms * 1e-3) / 1e12, maxerr);
  return 0;
}
