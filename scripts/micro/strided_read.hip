// How fast can a [N, D] fp32 tensor be read when every wavefront fetches, per step, a PIECE of `piece` bytes from each of
// `rows` different rows (the access pattern of the node-side GEMM operand: K slab x d floats per atom) -- against a plain
// streaming read of the same tensor?  Build: hipcc --offload-arch=gfx950 -O3 strided_read.hip -o strided_read.out
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

// each wave: rows_per_wave rows starting at row0; loops over the row in pieces of `piece` bytes (lanes cover
// rows_per_wave * piece / 16 float4s per step, several steps if more than 64 float4)
__global__ __launch_bounds__(256) void k_pieces(const float4* __restrict__ x, int64_t row_f4, int rows_per_wave, int piece_f4,
                                                int64_t n_rows, float* __restrict__ sink, int inflight) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t row0 = wave * rows_per_wave;
  if (row0 >= n_rows) return;
  float acc = 0.f;
  const int per_step = rows_per_wave * piece_f4;  // float4 per step over all rows
  const int steps = (int)(row_f4 / piece_f4);
  for (int s = 0; s < steps; s += inflight) {
    float4 v[8][8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (q >= inflight || s + q >= steps) break;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int idx = lane + 64 * i;
        if (idx < per_step) {
          const int r = idx / piece_f4, o = idx - r * piece_f4;
          const int64_t row = row0 + r < n_rows ? row0 + r : n_rows - 1;
          v[q][i] = x[row * row_f4 + (int64_t)(s + q) * piece_f4 + o];
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (q >= inflight || s + q >= steps) break;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (lane + 64 * i < per_step) acc += v[q][i].x + v[q][i].w;
    }
  }
  if (acc == 12345.f) sink[0] = acc;
}

__global__ __launch_bounds__(256) void k_stream(const float4* __restrict__ x, int64_t n_f4, float* __restrict__ sink) {
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_f4; i += (int64_t)gridDim.x * 256) {
    const float4 v = x[i];
    acc += v.x + v.w;
  }
  if (acc == 12345.f) sink[0] = acc;
}

int main(int argc, char** argv) {
  const int64_t N = argc > 1 ? atoll(argv[1]) : 10125;
  const int64_t D = argc > 2 ? atoll(argv[2]) : 2240;
  const int64_t bytes = N * D * 4;
  float4* x; float* sink;
  hipMalloc(&x, bytes); hipMalloc(&sink, 4);
  hipMemset(x, 0, bytes);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  auto time = [&](auto launch) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 10;
  };
  float ms = time([&] { hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, 0, x, bytes / 16, sink); });
  printf("N=%lld D=%lld (%.0f MB)  stream: %.1f us  %.2f TB/s\n", (long long)N, (long long)D, bytes / 1e6, ms * 1e3, bytes / ms / 1e9);
  for (int rows : {32, 10, 6}) {
    for (int piece : {128, 256, 512, 1024}) {
      if ((D * 4) % piece) continue;
      for (int inflight : {1, 2, 4}) {
        if ((int64_t)rows * piece / 16 > 512) continue;
        const int64_t waves = (N + rows - 1) / rows;
        ms = time([&] { hipLaunchKernelGGL(k_pieces, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, 0, x, D / 4, rows, piece / 16, N, sink, inflight); });
        printf("rows/wave %2d  piece %4d B  steps in flight %d: %.1f us  %.2f TB/s\n", rows, piece, inflight, ms * 1e3, bytes / ms / 1e9);
      }
    }
  }
  return 0;
}
