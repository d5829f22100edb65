// Write / read / copy bandwidth of the access shapes the pair-centric backward uses (round 6 study).
//   hipcc --offload-arch=gfx950 -O3 store_bw.hip -o store_bw.out ;  ./store_bw.out [MiB]
// Every wavefront walks rows of ROWB bytes (a grad_w row is 2816 B): row r of wave w is (w + k * waves); a store instruction
// covers 64 lanes x WIDTH bytes.  Reported: GB/s of payload per kernel form.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int W> struct Vec;
template <> struct Vec<1> { typedef float T; };
template <> struct Vec<4> { typedef float4 T; };

// mode bits: 1 = write, 2 = read, 4 = nontemporal stores
template <int W, int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ src, float* __restrict__ dst, size_t n_floats, float* sink) {
  typedef typename Vec<W>::T V;
  const size_t nv = n_floats / W;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  const V* s = reinterpret_cast<const V*>(src);
  V* d = reinterpret_cast<V*>(dst);
  float acc = 0.f;
  for (size_t i = tid; i < nv; i += stride) {
    V v;
    if (MODE & 2) { v = s[i]; if constexpr (W == 1) acc += v; else acc += v.x + v.w; }
    else { if constexpr (W == 1) v = (float)i; else v = make_float4((float)i, 1.f, 2.f, 3.f); }
    if (MODE & 1) {
      if (MODE & 4) {
        if constexpr (W == 1) __builtin_nontemporal_store(v, &d[i]);
        else { float* p = reinterpret_cast<float*>(&d[i]); typedef float f4 __attribute__((ext_vector_type(4))); f4 q = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(q, reinterpret_cast<f4*>(p)); }
      } else d[i] = v;
    }
  }
  if ((MODE & 2) && !(MODE & 1) && acc == 12345.678f) *sink = acc;
}

template <int W, int MODE>
static void run(const char* name, const float* src, float* dst, size_t n, float* sink, int blocks) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<W, MODE>), dim3(blocks), dim3(256), 0, 0, src, dst, n, sink);
  CK(hipEventRecord(e0, 0));
  const int reps = 10;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<W, MODE>), dim3(blocks), dim3(256), 0, 0, src, dst, n, sink);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  const double bytes = (double)n * 4 * (((MODE & 1) ? 1 : 0) + ((MODE & 2) ? 1 : 0));
  printf("%-34s blocks %6d  %8.1f us  %7.2f TB/s\n", name, blocks, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
  const size_t mib = argc > 1 ? atoi(argv[1]) : 1024;
  const size_t n = mib * 1024 * 1024 / 4;
  float *src, *dst, *sink;
  CK(hipMalloc(&src, n * 4)); CK(hipMalloc(&dst, n * 4)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(src, 1, n * 4)); CK(hipMemset(dst, 0, n * 4));
  for (int blocks : {2048, 8192, 65536}) {
    run<1, 1>("write dword/lane", src, dst, n, sink, blocks);
    run<1, 5>("write dword/lane nt", src, dst, n, sink, blocks);
    run<4, 1>("write dwordx4/lane", src, dst, n, sink, blocks);
    run<4, 5>("write dwordx4/lane nt", src, dst, n, sink, blocks);
    run<1, 2>("read dword/lane", src, dst, n, sink, blocks);
    run<4, 2>("read dwordx4/lane", src, dst, n, sink, blocks);
    run<1, 3>("copy dword/lane", src, dst, n, sink, blocks);
    run<4, 3>("copy dwordx4/lane", src, dst, n, sink, blocks);
    run<4, 7>("copy dwordx4/lane nt stores", src, dst, n, sink, blocks);
  }
  {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMemsetAsync(dst, 0, n * 4, 0));
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < 10; ++r) CK(hipMemsetAsync(dst, 0, n * 4, 0));
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    printf("%-34s                %8.1f us  %7.2f TB/s\n", "hipMemsetAsync", ms * 1e3, (double)n * 4 / (ms * 1e-3) / 1e12);
  }
  return 0;
}
