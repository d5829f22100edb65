// Edge displacement vectors and their adjoint (the position/cell leg of the force and virial backward).
//
// Replaces with_edge_vectors_ (nequip/nn/utils.py:68-118):
//   edge_vec = pos[edge_index[1]] - pos[edge_index[0]] (+ edge_cell_shift @ cell[frame of edge_index[0]])
// and the autograd of those index_select / baddbmm ops, which on the GPU become float64 atomic index_add
// kernels over E rows (1.5 ms per evaluation on the 10k-atom water box, profiles/).  The adjoint here is
// atomics-free and deterministic: one wavefront per atom walks the atom's edge lists in both CSRs
// (as centre: -g, as neighbour: +g) with lanes over edges and a wave reduction.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "plan.h"

namespace nqa {

__global__ __launch_bounds__(256) void edge_vectors_fwd_kernel(const double* __restrict__ pos,
                                                               const int64_t* __restrict__ dst,
                                                               const int64_t* __restrict__ src,
                                                               const double* __restrict__ shift,
                                                               const double* __restrict__ cell,
                                                               const int64_t* __restrict__ batch, int64_t E,
                                                               double* __restrict__ vec) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int64_t d = dst[e], s = src[e];
  double vx = pos[3 * s + 0] - pos[3 * d + 0];
  double vy = pos[3 * s + 1] - pos[3 * d + 1];
  double vz = pos[3 * s + 2] - pos[3 * d + 2];
  if (cell != nullptr) {
    const double* __restrict__ c = cell + (batch ? 9 * batch[d] : 0);
    const double s0 = shift[3 * e + 0], s1 = shift[3 * e + 1], s2 = shift[3 * e + 2];
    // row-vector convention (ASE): shift @ cell
    vx += s0 * c[0] + s1 * c[3] + s2 * c[6];
    vy += s0 * c[1] + s1 * c[4] + s2 * c[7];
    vz += s0 * c[2] + s1 * c[5] + s2 * c[8];
  }
  vec[3 * e + 0] = vx;
  vec[3 * e + 1] = vy;
  vec[3 * e + 2] = vz;
}

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// g_pos[n] = sum_{e: src(e)=n} g[e] - sum_{e: dst(e)=n} g[e];  cell_part[n] = sum_{e: dst(e)=n} shift[e]^T g[e]
__global__ __launch_bounds__(256) void edge_vectors_bwd_kernel(const double* __restrict__ g,
                                                               const double* __restrict__ shift,
                                                               const int32_t* __restrict__ rowptr_dst,
                                                               const int32_t* __restrict__ eid_dst,
                                                               const int32_t* __restrict__ rowptr_src,
                                                               const int32_t* __restrict__ eid_src, int64_t N,
                                                               double sign, double* __restrict__ g_pos,
                                                               double* __restrict__ cell_part) {
  const int lane = threadIdx.x & 63;
  const int64_t n = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (n >= N) return;
  double ax = 0.0, ay = 0.0, az = 0.0;
  double m[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) m[i] = 0.0;
  for (int idx = rowptr_dst[n] + lane; idx < rowptr_dst[n + 1]; idx += 64) {
    const int64_t e = eid_dst[idx];
    const double gx = g[3 * e + 0], gy = g[3 * e + 1], gz = g[3 * e + 2];
    ax -= gx;
    ay -= gy;
    az -= gz;
    if (cell_part != nullptr) {
      const double s0 = shift[3 * e + 0], s1 = shift[3 * e + 1], s2 = shift[3 * e + 2];
      m[0] += s0 * gx; m[1] += s0 * gy; m[2] += s0 * gz;
      m[3] += s1 * gx; m[4] += s1 * gy; m[5] += s1 * gz;
      m[6] += s2 * gx; m[7] += s2 * gy; m[8] += s2 * gz;
    }
  }
  for (int idx = rowptr_src[n] + lane; idx < rowptr_src[n + 1]; idx += 64) {
    const int64_t e = eid_src[idx];
    ax += g[3 * e + 0];
    ay += g[3 * e + 1];
    az += g[3 * e + 2];
  }
  ax = wave_sum_f64(ax);
  ay = wave_sum_f64(ay);
  az = wave_sum_f64(az);
  if (cell_part != nullptr) {
#pragma unroll
    for (int i = 0; i < 9; ++i) m[i] = wave_sum_f64(m[i]);
  }
  if (lane == 0) {
    g_pos[3 * n + 0] = sign * ax;
    g_pos[3 * n + 1] = sign * ay;
    g_pos[3 * n + 2] = sign * az;
    if (cell_part != nullptr) {
#pragma unroll
      for (int i = 0; i < 9; ++i) cell_part[9 * n + i] = m[i];
    }
  }
}

// One workgroup per frame: m = sum over the frame's atoms of part[n] (ordered tree reduction), virial = -sym(m),
// stress = sym(m) / |det cell|.  Replaces the tail of ForceStressOutput.forward (nequip/nn/grad_output.py:222-271).
// (round 6: 1024 threads per frame -- the 10 125-atom box kept one 256-thread workgroup busy for 19 us at the very end of
// the step -- wavefront sums by DPP-free shuffles in a fixed order, then 16 partial rows through LDS: deterministic)
__global__ __launch_bounds__(1024) void virial_finalize_kernel(const double* __restrict__ part,
                                                               const int64_t* __restrict__ batch,
                                                               const double* __restrict__ cell, int64_t N,
                                                               double* __restrict__ virial,
                                                               double* __restrict__ stress) {
  __shared__ double red[9][16];
  const int f = blockIdx.x, tid = threadIdx.x;
  double m[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) m[i] = 0.0;
  for (int64_t n = tid; n < N; n += 1024) {
    if (batch != nullptr && batch[n] != f) continue;
#pragma unroll
    for (int i = 0; i < 9; ++i) m[i] += part[9 * n + i];
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m[i] += __shfl_down(m[i], off, 64);
  }
  if ((tid & 63) == 0) {
#pragma unroll
    for (int i = 0; i < 9; ++i) red[i][tid >> 6] = m[i];
  }
  __syncthreads();
  if (tid < 9) {
    const int a = tid / 3, b = tid - 3 * a;
    double sab = 0.0, sba = 0.0;
    for (int w = 0; w < 16; ++w) {
      sab += red[3 * a + b][w];
      sba += red[3 * b + a][w];
    }
    const double sym = 0.5 * (sab + sba);
    virial[9 * f + tid] = -sym;
    if (stress != nullptr) {
      const double* __restrict__ c = cell + 9 * f;
      const double vol = fabs(c[0] * (c[4] * c[8] - c[5] * c[7]) - c[1] * (c[3] * c[8] - c[5] * c[6]) +
                              c[2] * (c[3] * c[7] - c[4] * c[6]));
      stress[9 * f + tid] = sym / vol;
    }
  }
}

// out[f, :] = sum over the atoms n of frame f of rows[n, :] (K <= 16 columns): one workgroup per frame, ordered tree
// reduction -- the deterministic replacement of `zeros(B, K).index_add_(0, batch, rows)`, whose 8192 x 9 float64 atomics
// on 32 x 9 addresses take 38 us.
__global__ __launch_bounds__(256) void frame_sum_kernel(const double* __restrict__ rows,
                                                        const int64_t* __restrict__ batch, int64_t N, int K,
                                                        double* __restrict__ out) {
  __shared__ double red[16][256];
  const int f = blockIdx.x, tid = threadIdx.x;
  double m[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) m[i] = 0.0;
  for (int64_t n = tid; n < N; n += 256) {
    if (batch != nullptr && batch[n] != f) continue;
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (i < K) m[i] += rows[(int64_t)K * n + i];
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) red[i][tid] = m[i];
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) {
#pragma unroll
      for (int i = 0; i < 16; ++i) red[i][tid] += red[i][tid + off];
    }
    __syncthreads();
  }
  if (tid < K) out[(int64_t)K * f + tid] = red[tid][0];
}

}  // namespace nqa

using namespace nqa;

extern "C" {

int nqa_edge_vectors_fwd(const double* pos, const int64_t* edge_dst, const int64_t* edge_src,
                         const double* edge_cell_shift, const double* cell, const int64_t* batch, int64_t num_edges,
                         double* edge_vec, nqa_stream stream) {
  // (empty tensors have NULL data pointers: operands are only required when there are edges)
  if (num_edges < 0 || (num_edges > 0 && (!pos || !edge_dst || !edge_src || !edge_vec)) ||
      (num_edges > 0 && cell != nullptr && edge_cell_shift == nullptr)) {
    set_error("nqa_edge_vectors_fwd: invalid argument");
    return NQA_ERR_INVALID;
  }
  if (num_edges == 0) return NQA_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(edge_vectors_fwd_kernel, dim3((unsigned)((num_edges + 255) / 256)), dim3(256), 0, s, pos,
                     edge_dst, edge_src, edge_cell_shift, cell, batch, num_edges, edge_vec);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_error(std::string("nqa_edge_vectors_fwd: ") + hipGetErrorString(err));
    return NQA_ERR_LAUNCH;
  }
  return NQA_OK;
}

int nqa_edge_vectors_bwd(const double* g_edge_vec, const double* edge_cell_shift, const int32_t* rowptr_dst,
                         const int32_t* edge_id_dst, const int32_t* rowptr_src, const int32_t* edge_id_src,
                         int64_t num_nodes, double sign, double* g_pos, double* g_cell_per_node, nqa_stream stream) {
  // (g_edge_vec / edge_cell_shift are only dereferenced inside non-empty edge rows: NULL is legal for a graph without
  // edges, where empty tensors have NULL data pointers)
  if (num_nodes < 0 || (num_nodes > 0 && (!g_pos || !rowptr_dst || !rowptr_src))) {
    set_error("nqa_edge_vectors_bwd: invalid argument");
    return NQA_ERR_INVALID;
  }
  if (num_nodes == 0) return NQA_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(edge_vectors_bwd_kernel, dim3((unsigned)((num_nodes + 3) / 4)), dim3(256), 0, s, g_edge_vec,
                     edge_cell_shift, rowptr_dst, edge_id_dst, rowptr_src, edge_id_src, num_nodes, sign, g_pos,
                     g_cell_per_node);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_error(std::string("nqa_edge_vectors_bwd: ") + hipGetErrorString(err));
    return NQA_ERR_LAUNCH;
  }
  return NQA_OK;
}

int nqa_virial_finalize(const double* per_atom, const int64_t* batch, const double* cell, int64_t num_nodes,
                        int64_t num_frames, double* virial, double* stress, nqa_stream stream) {
  if (num_nodes < 0 || num_frames < 0 || (num_frames > 0 && !virial) || (num_nodes > 0 && !per_atom) ||
      (stress != nullptr && cell == nullptr) || (num_frames > 1 && batch == nullptr)) {
    set_error("nqa_virial_finalize: invalid argument");
    return NQA_ERR_INVALID;
  }
  if (num_frames == 0) return NQA_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(virial_finalize_kernel, dim3((unsigned)num_frames), dim3(1024), 0, s, per_atom,
                     num_frames > 1 ? batch : nullptr, cell, num_nodes, virial, stress);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_error(std::string("nqa_virial_finalize: ") + hipGetErrorString(err));
    return NQA_ERR_LAUNCH;
  }
  return NQA_OK;
}

int nqa_frame_sum(const double* rows, const int64_t* batch, int64_t num_nodes, int32_t width, int64_t num_frames,
                  double* out, nqa_stream stream) {
  if (num_nodes < 0 || num_frames < 0 || width < 1 || width > 16 || (num_frames > 0 && !out) ||
      (num_nodes > 0 && !rows) || (num_frames > 1 && batch == nullptr)) {
    set_error("nqa_frame_sum: invalid argument (1 <= width <= 16; batch is required for more than one frame)");
    return NQA_ERR_INVALID;
  }
  if (num_frames == 0) return NQA_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(frame_sum_kernel, dim3((unsigned)num_frames), dim3(256), 0, s, rows, batch, num_nodes, (int)width,
                     out);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_error(std::string("nqa_frame_sum: ") + hipGetErrorString(err));
    return NQA_ERR_LAUNCH;
  }
  return NQA_OK;
}

}  // extern "C"
