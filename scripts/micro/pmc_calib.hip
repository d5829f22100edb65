// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of THIS repo's kernels.
// MI355X_MICROARCH.md documents "FETCH_SIZE = half the bytes" for 16 B/lane streaming reads only and says to calibrate other
// widths on a known byte count.  Every kernel below moves exactly `bytes` (printed) between HBM and the CUs, on buffers
// four times the 256 MiB Infinity Cache, so counter x 1024 / bytes is the factor scripts/summarize_profile.py needs:
//   calib_read16        float4 per lane, streaming                       (radial MLP backward's gradient rows, node rows)
//   calib_read4_rows    4 B per lane, one 256 B row segment per wave-load, rows of `W` floats gathered in random order
//                       (the tensor-product kernels: lane = channel, weight row of the edge)
//   calib_write4_rows   the same pattern with nontemporal 4 B stores     (grad_w / per-pair grad_x rows)
//   calib_write16_nt    nontemporal float4 stores of full 128 B lines    (radial MLP forward's weight rows)
//   calib_write4_plain  plain 4 B stores, row pattern
// Build: hipcc --offload-arch=gfx950 -O3 pmc_calib.hip -o pmc_calib.out     Run: scripts/pmc_calibrate.py (two --pmc passes).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                         \
  do {                                                                                   \
    hipError_t e_ = (x);                                                                 \
    if (e_ != hipSuccess) {                                                              \
      fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));                     \
      exit(1);                                                                           \
    }                                                                                    \
  } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void calib_read16(const float4* __restrict__ x, int64_t n4, float* __restrict__ sink) {
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 v = x[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 12345.678f) sink[0] = acc;
}

// one wavefront per "edge": row r = perm(edge) of [rows, W] floats, read as W / 64 wave-loads of 4 B per lane
__device__ __forceinline__ int64_t calib_perm(int64_t e, int64_t rows) { return (e * 2654435761LL + 12345) % rows; }

__global__ __launch_bounds__(256) void calib_read4_rows(const float* __restrict__ x, int64_t rows, int W, float* __restrict__ sink) {
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
  for (int64_t e = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); e < rows; e += (int64_t)gridDim.x * 4) {
    const float* __restrict__ r = x + calib_perm(e, rows) * W;
    for (int c = 0; c < W; c += 64) acc += r[c + lane];
  }
  if (acc == 12345.678f) sink[0] = acc;
}

template <bool NT>
__global__ __launch_bounds__(256) void calib_write4_rows(float* __restrict__ x, int64_t rows, int W) {
  const int lane = threadIdx.x & 63;
  for (int64_t e = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); e < rows; e += (int64_t)gridDim.x * 4) {
    float* __restrict__ r = x + calib_perm(e, rows) * W;
    for (int c = 0; c < W; c += 64) {
      if (NT)
        __builtin_nontemporal_store((float)(e + c), r + c + lane);
      else
        r[c + lane] = (float)(e + c);
    }
  }
}

// 8 lanes x 16 B = one 128 B line per row and instruction, 8 rows per instruction, row stride W floats (radial MLP emit())
__global__ __launch_bounds__(256) void calib_write16_nt(float* __restrict__ x, int64_t rows, int W) {
  const int lane = threadIdx.x & 63, c4 = lane & 7, rsub = lane >> 3;
  const int64_t nblk = rows / 32;
  for (int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); b < nblk; b += (int64_t)gridDim.x * 4)
    for (int n0 = 0; n0 < W; n0 += 32)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_nontemporal_store((v4f){1.f, 2.f, 3.f, (float)n0}, reinterpret_cast<v4f*>(x + (b * 32 + 8 * i + rsub) * W + n0 + 4 * c4));
}

int main(int argc, char** argv) {
  const int W = 704;
  const int64_t rows = (int64_t)(1 << 20) / W * 365;  // ~1.0 GiB of fp32 rows, a multiple of 32 rows
  const int64_t rows32 = rows / 32 * 32;
  const int64_t n = rows32 * W;
  const double bytes = 4.0 * n;
  float *x, *sink;
  CHECK(hipMalloc(&x, n * sizeof(float)));
  CHECK(hipMalloc(&sink, 256));
  CHECK(hipMemset(x, 0, n * sizeof(float)));
  const int grid = 256 * 8;
  for (int rep = 0; rep < 3; ++rep) {
    calib_read16<<<grid, 256>>>(reinterpret_cast<const float4*>(x), n / 4, sink);
    calib_read4_rows<<<grid, 256>>>(x, rows32, W, sink);
    calib_write4_rows<true><<<grid, 256>>>(x, rows32, W);
    calib_write4_rows<false><<<grid, 256>>>(x, rows32, W);
    calib_write16_nt<<<grid, 256>>>(x, rows32, W);
  }
  CHECK(hipDeviceSynchronize());
  printf("{\"bytes_per_kernel\": %.0f, \"rows\": %lld, \"W\": %d}\n", bytes, (long long)rows32, W);
  return 0;
}
