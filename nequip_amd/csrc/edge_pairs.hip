// Reverse-edge pairing of a neighbour list, on device.
//
// The radial network of an interaction block (InteractionBlock.edge_mlp, nequip/nn/interaction_block.py:119-127,190-192)
// maps the edge embedding -- a function of the edge LENGTH only (BesselEdgeLengthEncoding + PolynomialCutoff,
// nequip/nn/embedding/_edge.py:136-150) -- to the per-edge tensor-product weights.  A neighbour list built with one
// cutoff contains every interaction twice, as (i <- j, S) and (j <- i, -S), with bitwise identical lengths, so the
// reference evaluates that MLP (the only dense GEMM on the edge side, 182 kFLOP/edge) twice per pair.  This kernel family
// finds the pairs so that the MLP runs once per pair and the tensor-product kernels read the shared row through an index
// (nqa_tp_scatter_*_paired):
//   weight_rows[e] = p      for the representative edge of pair p (dst < src, or dst == src with a "positive" shift)
//                  = p + P  for its reverse               (P = E / 2 pairs)
//   rep_edge[p]    = the representative edge
// Method: canonical 64-bit key (min(i,j), max(i,j), shift of the canonical orientation) per edge, radix sort of
// (key, edge id), then every even sorted position must hold exactly two edges with the same key and opposite
// orientation.  Anything else (odd E, a missing reverse edge, duplicates, shifts outside [-8, 7], > 2^26 atoms) clears
// the `ok` flag and the caller keeps the per-edge evaluation.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <cstdint>
#include <string>

#include "plan.h"

namespace nqa {

static int64_t ep_align256(int64_t v) { return (v + 255) / 256 * 256; }

struct PairKey {
  uint64_t key;
  int rep;  // 1: this edge has the canonical orientation
  int bad;
};

template <typename ST>
__device__ __forceinline__ PairKey pair_key(const int64_t* __restrict__ dst, const int64_t* __restrict__ src,
                                            const ST* __restrict__ shift, int64_t e, int64_t N) {
  PairKey r;
  const int64_t i = dst[e], j = src[e];
  int sx = 0, sy = 0, sz = 0;
  if (shift != nullptr) {
    sx = (int)lrint((double)shift[3 * e + 0]);
    sy = (int)lrint((double)shift[3 * e + 1]);
    sz = (int)lrint((double)shift[3 * e + 2]);
  }
  r.bad = (i < 0 || j < 0 || i >= N || j >= N || i >= (1 << 26) || j >= (1 << 26) || sx < -8 || sx > 7 || sy < -8 ||
           sy > 7 || sz < -8 || sz > 7)
              ? 1
              : 0;
  // canonical orientation: dst < src; self images by the sign of the first non-zero shift component
  bool rep;
  if (i != j) rep = i < j;
  else rep = sx > 0 || (sx == 0 && (sy > 0 || (sy == 0 && sz > 0)));
  if (i == j && sx == 0 && sy == 0 && sz == 0) r.bad = 1;  // a self edge without a shift has no partner
  const int64_t lo = rep ? i : j, hi = rep ? j : i;
  if (!rep) { sx = -sx; sy = -sy; sz = -sz; }
  if (sx < -8 || sx > 7 || sy < -8 || sy > 7 || sz < -8 || sz > 7) r.bad = 1;
  const uint64_t sc = (uint64_t)((sx + 8) & 15) << 8 | (uint64_t)((sy + 8) & 15) << 4 | (uint64_t)((sz + 8) & 15);
  r.key = ((uint64_t)(lo & ((1 << 26) - 1)) << 38) | ((uint64_t)(hi & ((1 << 26) - 1)) << 12) | sc;
  r.rep = rep ? 1 : 0;
  return r;
}

template <typename ST>
__global__ __launch_bounds__(256) void edge_pairs_key_kernel(const int64_t* __restrict__ dst,
                                                              const int64_t* __restrict__ src,
                                                              const ST* __restrict__ shift, int64_t E, int64_t N,
                                                              uint64_t* __restrict__ keys, int32_t* __restrict__ vals,
                                                              int32_t* __restrict__ ok) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const PairKey k = pair_key(dst, src, shift, e, N);
  if (k.bad) atomicAnd(ok, 0);
  keys[e] = k.key;
  vals[e] = (int32_t)e;
}

template <typename ST>
__global__ __launch_bounds__(256) void edge_pairs_assign_kernel(const int64_t* __restrict__ dst,
                                                                 const int64_t* __restrict__ src,
                                                                 const ST* __restrict__ shift, int64_t E, int64_t N,
                                                                 const uint64_t* __restrict__ keys_sorted,
                                                                 const int32_t* __restrict__ vals_sorted,
                                                                 int32_t* __restrict__ weight_rows,
                                                                 int64_t* __restrict__ rep_edge,
                                                                 int32_t* __restrict__ ok) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // pair index
  const int64_t P = E / 2;
  if (p >= P) return;
  const uint64_t k0 = keys_sorted[2 * p], k1 = keys_sorted[2 * p + 1];
  const int32_t e0 = vals_sorted[2 * p], e1 = vals_sorted[2 * p + 1];
  bool good = k0 == k1;
  if (2 * p + 2 < E && keys_sorted[2 * p + 2] == k0) good = false;  // more than two edges with this key
  const int r0 = pair_key(dst, src, shift, (int64_t)e0, N).rep;
  const int r1 = pair_key(dst, src, shift, (int64_t)e1, N).rep;
  if (r0 == r1) good = false;  // duplicates instead of a reverse edge
  if (!good) {
    atomicAnd(ok, 0);
    return;
  }
  const int32_t rep = r0 ? e0 : e1, other = r0 ? e1 : e0;
  weight_rows[rep] = (int32_t)p;
  weight_rows[other] = (int32_t)(p + P);
  rep_edge[p] = (int64_t)rep;
}

// out[p, :] = src[rep_edge[p], :]  (rows of `width` 32-bit words)
__global__ __launch_bounds__(256) void pair_gather_kernel(const uint32_t* __restrict__ src,
                                                          const int64_t* __restrict__ rep_edge, int64_t P, int width,
                                                          uint32_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P * width) return;
  const int64_t p = i / width;
  const int c = (int)(i - p * width);
  out[i] = src[rep_edge[p] * width + c];
}

// adjoint: out[e, :] = weight_rows[e] < P ? g[weight_rows[e], :] : 0   (every row of out is written: no memset)
__global__ __launch_bounds__(256) void pair_expand_kernel(const uint32_t* __restrict__ g,
                                                          const int32_t* __restrict__ weight_rows, int64_t E, int64_t P,
                                                          int width, uint32_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= E * width) return;
  const int64_t e = i / width;
  const int c = (int)(i - e * width);
  const int32_t r = weight_rows[e];
  out[i] = r < P ? g[(int64_t)r * width + c] : 0u;
}

// ---- owner lists of the pair-centric backward, from the pairing and the dst-CSR (no sort) ----------------------------------
// Every directed edge e (dst = n) is either the owner-side edge of its pair (`edge_in`: n owns the pair) or the other-side
// edge (`edge_out`: n is the pair's other node).  Owner of {i <- j, j <- i}: i when (i < j) xor (i + j odd); a self-image pair
// (i == j) is owned through its representative edge.  So both lists of a node are subsequences of its CSR row.
__device__ __forceinline__ bool ep_is_in(int i, int j, int32_t row, int32_t P) {
  return i != j ? ((i < j) != (((i + j) & 1) == 1)) : row < P;
}

__global__ __launch_bounds__(256) void pair_reverse_edge_kernel(const int32_t* __restrict__ rows, int64_t E, int32_t P,
                                                                int32_t* __restrict__ rev_edge) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int32_t r = rows[e];
  if (r >= P) rev_edge[r - P] = (int32_t)e;
}

__global__ __launch_bounds__(256) void pair_owner_count_kernel(const int32_t* __restrict__ rowptr,
                                                               const int32_t* __restrict__ edge_id,
                                                               const int32_t* __restrict__ src_sorted,
                                                               const int32_t* __restrict__ rows, int64_t N, int32_t P,
                                                               int32_t* __restrict__ cnt_own, int32_t* __restrict__ cnt_oth) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n > N) return;
  int own = 0, oth = 0;
  if (n < N) {
    for (int32_t k = rowptr[n]; k < rowptr[n + 1]; ++k) {
      const bool in = ep_is_in((int)n, src_sorted[k], rows[edge_id[k]], P);
      own += in ? 1 : 0;
      oth += in ? 0 : 1;
    }
  }
  cnt_own[n] = own;  // (entry N = 0: the exclusive scan over N + 1 entries ends in the total)
  cnt_oth[n] = oth;
}

__global__ __launch_bounds__(256) void pair_owner_fill_kernel(const int32_t* __restrict__ rowptr,
                                                              const int32_t* __restrict__ edge_id,
                                                              const int32_t* __restrict__ src_sorted,
                                                              const int32_t* __restrict__ rows,
                                                              const int64_t* __restrict__ rep_edge,
                                                              const int32_t* __restrict__ rev_edge, int64_t N, int32_t P,
                                                              const int32_t* __restrict__ owner_rowptr,
                                                              int32_t* __restrict__ pair_other, int32_t* __restrict__ pair_row,
                                                              int32_t* __restrict__ edge_in, int32_t* __restrict__ edge_out,
                                                              int32_t* __restrict__ slot_of_pair) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  int32_t s = owner_rowptr[n];
  for (int32_t k = rowptr[n]; k < rowptr[n + 1]; ++k) {
    const int32_t e = edge_id[k], j = src_sorted[k], r = rows[e];
    if (!ep_is_in((int)n, j, r, P)) continue;
    const int32_t p = r < P ? r : r - P;
    pair_other[s] = j;
    pair_row[s] = p;
    edge_in[s] = e;
    edge_out[s] = r < P ? rev_edge[p] : (int32_t)rep_edge[p];
    slot_of_pair[p] = s;
    ++s;
  }
}

__global__ __launch_bounds__(256) void pair_other_fill_kernel(const int32_t* __restrict__ rowptr,
                                                              const int32_t* __restrict__ edge_id,
                                                              const int32_t* __restrict__ src_sorted,
                                                              const int32_t* __restrict__ rows, int64_t N, int32_t P,
                                                              const int32_t* __restrict__ other_rowptr,
                                                              const int32_t* __restrict__ slot_of_pair,
                                                              int32_t* __restrict__ other_slot) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  int32_t t = other_rowptr[n];
  for (int32_t k = rowptr[n]; k < rowptr[n + 1]; ++k) {
    const int32_t r = rows[edge_id[k]];
    if (ep_is_in((int)n, src_sorted[k], r, P)) continue;
    other_slot[t++] = slot_of_pair[r < P ? r : r - P];
  }
}

static size_t ep_scan_bytes(int64_t N) {
  size_t bytes = 0;
  (void)rocprim::exclusive_scan(nullptr, bytes, (const int32_t*)nullptr, (int32_t*)nullptr, (int32_t)0, (size_t)(N + 1),
                                rocprim::plus<int32_t>(), (hipStream_t)0);
  return bytes;
}

static size_t ep_cub_bytes(int64_t E) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const int32_t*)nullptr,
                                  (int32_t*)nullptr, (unsigned int)E, 0u, 64u, (hipStream_t)0);
  return bytes;
}

}  // namespace nqa

using namespace nqa;

extern "C" {

int64_t nqa_edge_pairs_workspace_bytes(int64_t num_edges) {
  if (num_edges < 0 || num_edges > 2147483647LL) return -1;
  const int64_t E = num_edges > 0 ? num_edges : 1;
  return 2 * ep_align256(E * 8) + 2 * ep_align256(E * 4) + ep_align256((int64_t)ep_cub_bytes(E));
}

int nqa_edge_pairs(const int64_t* edge_dst, const int64_t* edge_src, const void* edge_cell_shift, int32_t shift_dtype,
                   int64_t num_edges, int64_t num_nodes, void* workspace, int64_t workspace_bytes,
                   int32_t* weight_rows, int64_t* rep_edge, int32_t* ok, nqa_stream stream) {
  if (num_edges < 0 || num_nodes < 0 || !ok || (num_edges > 0 && (!edge_dst || !edge_src || !weight_rows || !rep_edge)) ||
      (edge_cell_shift && shift_dtype != NQA_F32 && shift_dtype != NQA_F64)) {
    set_error("nqa_edge_pairs: invalid argument");
    return NQA_ERR_INVALID;
  }
  const int64_t need = nqa_edge_pairs_workspace_bytes(num_edges);
  if (need < 0) {
    set_error("nqa_edge_pairs: too many edges");
    return NQA_ERR_UNSUPPORTED;
  }
  if (num_edges > 0 && (!workspace || workspace_bytes < need)) {
    set_error("nqa_edge_pairs: workspace missing or too small");
    return NQA_ERR_WORKSPACE;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  // ok = 1 unless E is odd or empty; the kernels clear it on any violation
  const int32_t init = (num_edges > 0 && (num_edges % 2) == 0) ? 1 : 0;
  if (hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ok), init, 1, s) != hipSuccess) {
    set_error("nqa_edge_pairs: flag initialisation failed");
    return NQA_ERR_LAUNCH;
  }
  if (!init) return NQA_OK;
  const int64_t E = num_edges;
  char* p = static_cast<char*>(workspace);
  uint64_t* keys = reinterpret_cast<uint64_t*>(p);
  p += ep_align256(E * 8);
  uint64_t* keys_sorted = reinterpret_cast<uint64_t*>(p);
  p += ep_align256(E * 8);
  int32_t* vals = reinterpret_cast<int32_t*>(p);
  p += ep_align256(E * 4);
  int32_t* vals_sorted = reinterpret_cast<int32_t*>(p);
  p += ep_align256(E * 4);
  size_t cub_bytes = ep_cub_bytes(E);
  const unsigned ge = (unsigned)((E + 255) / 256), gp = (unsigned)((E / 2 + 255) / 256);
#define NQA_EP_RUN(ST)                                                                                          \
  {                                                                                                            \
    const ST* sh = static_cast<const ST*>(edge_cell_shift);                                                     \
    hipLaunchKernelGGL(edge_pairs_key_kernel<ST>, dim3(ge), dim3(256), 0, s, edge_dst, edge_src, sh, E,         \
                       num_nodes, keys, vals, ok);                                                              \
    if (rocprim::radix_sort_pairs(p, cub_bytes, keys, keys_sorted, vals, vals_sorted, (unsigned int)E, 0u, 64u,  \
                                  s) != hipSuccess) {                                                           \
      set_error("nqa_edge_pairs: radix sort failed");                                                           \
      return NQA_ERR_LAUNCH;                                                                                    \
    }                                                                                                          \
    hipLaunchKernelGGL(edge_pairs_assign_kernel<ST>, dim3(gp), dim3(256), 0, s, edge_dst, edge_src, sh, E,      \
                       num_nodes, keys_sorted, vals_sorted, weight_rows, rep_edge, ok);                         \
  }
  if (edge_cell_shift && shift_dtype == NQA_F32) NQA_EP_RUN(float) else NQA_EP_RUN(double)
#undef NQA_EP_RUN
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_error(std::string("nqa_edge_pairs: ") + hipGetErrorString(err));
    return NQA_ERR_LAUNCH;
  }
  return NQA_OK;
}

int64_t nqa_pair_owner_workspace_bytes(int64_t num_edges, int64_t num_nodes) {
  if (num_edges < 0 || num_nodes < 0 || num_edges > 2147483647LL || num_nodes > 2147483646LL) return -1;
  const int64_t P = num_edges / 2 > 0 ? num_edges / 2 : 1;
  return 2 * ep_align256(P * 4) + 2 * ep_align256((num_nodes + 1) * 4) + ep_align256((int64_t)ep_scan_bytes(num_nodes));
}

int nqa_pair_owner_lists(const int32_t* weight_rows, const int64_t* rep_edge, const int32_t* rowptr_dst,
                         const int32_t* edge_id_dst, const int32_t* src_sorted, int64_t num_edges, int64_t num_nodes,
                         void* workspace, int64_t workspace_bytes, int32_t* owner_rowptr, int32_t* pair_other,
                         int32_t* pair_row, int32_t* pair_edge_in, int32_t* pair_edge_out, int32_t* other_rowptr,
                         int32_t* other_slot, nqa_stream stream) {
  if (num_edges < 0 || num_nodes < 0 || (num_edges % 2) != 0 || !owner_rowptr || !other_rowptr ||
      (num_edges > 0 && (!weight_rows || !rep_edge || !rowptr_dst || !edge_id_dst || !src_sorted || !pair_other || !pair_row ||
                         !pair_edge_in || !pair_edge_out || !other_slot))) {
    set_error("nqa_pair_owner_lists: invalid argument (a paired list has an even number of edges)");
    return NQA_ERR_INVALID;
  }
  const int64_t need = nqa_pair_owner_workspace_bytes(num_edges, num_nodes);
  if (need < 0) {
    set_error("nqa_pair_owner_lists: sizes beyond the int32 index range");
    return NQA_ERR_UNSUPPORTED;
  }
  if (!workspace || workspace_bytes < need) {
    set_error("nqa_pair_owner_lists: workspace missing or too small");
    return NQA_ERR_WORKSPACE;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t E = num_edges, N = num_nodes;
  const int32_t P = (int32_t)(E / 2);
  char* p = static_cast<char*>(workspace);
  int32_t* rev_edge = reinterpret_cast<int32_t*>(p);
  p += ep_align256((P > 0 ? P : 1) * 4LL);
  int32_t* slot_of_pair = reinterpret_cast<int32_t*>(p);
  p += ep_align256((P > 0 ? P : 1) * 4LL);
  int32_t* cnt_own = reinterpret_cast<int32_t*>(p);
  p += ep_align256((N + 1) * 4);
  int32_t* cnt_oth = reinterpret_cast<int32_t*>(p);
  p += ep_align256((N + 1) * 4);
  size_t scan_bytes = ep_scan_bytes(N);
  const unsigned gn = (unsigned)((N + 1 + 255) / 256);
  if (E > 0)
    hipLaunchKernelGGL(pair_reverse_edge_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, s, weight_rows, E, P,
                       rev_edge);
  hipLaunchKernelGGL(pair_owner_count_kernel, dim3(gn), dim3(256), 0, s, rowptr_dst, edge_id_dst, src_sorted, weight_rows, N,
                     P, cnt_own, cnt_oth);
  if (rocprim::exclusive_scan(p, scan_bytes, cnt_own, owner_rowptr, (int32_t)0, (size_t)(N + 1), rocprim::plus<int32_t>(),
                              s) != hipSuccess ||
      rocprim::exclusive_scan(p, scan_bytes, cnt_oth, other_rowptr, (int32_t)0, (size_t)(N + 1), rocprim::plus<int32_t>(),
                              s) != hipSuccess) {
    set_error("nqa_pair_owner_lists: prefix sum failed");
    return NQA_ERR_LAUNCH;
  }
  if (N > 0 && E > 0) {
    hipLaunchKernelGGL(pair_owner_fill_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, rowptr_dst, edge_id_dst,
                       src_sorted, weight_rows, rep_edge, rev_edge, N, P, owner_rowptr, pair_other, pair_row, pair_edge_in,
                       pair_edge_out, slot_of_pair);
    hipLaunchKernelGGL(pair_other_fill_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, rowptr_dst, edge_id_dst,
                       src_sorted, weight_rows, N, P, other_rowptr, slot_of_pair, other_slot);
  }
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_error(std::string("nqa_pair_owner_lists: ") + hipGetErrorString(err));
    return NQA_ERR_LAUNCH;
  }
  return NQA_OK;
}

int nqa_pair_gather(const void* rows_in, const int64_t* rep_edge, int64_t num_pairs, int32_t width, void* rows_out,
                    nqa_stream stream) {
  if (num_pairs < 0 || width <= 0 || (num_pairs > 0 && (!rows_in || !rep_edge || !rows_out))) {
    set_error("nqa_pair_gather: invalid argument");
    return NQA_ERR_INVALID;
  }
  if (num_pairs == 0) return NQA_OK;
  const int64_t n = num_pairs * width;
  hipLaunchKernelGGL(pair_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const uint32_t*>(rows_in), rep_edge, num_pairs,
                     width, static_cast<uint32_t*>(rows_out));
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_error(std::string("nqa_pair_gather: ") + hipGetErrorString(err));
    return NQA_ERR_LAUNCH;
  }
  return NQA_OK;
}

int nqa_pair_expand(const void* pair_rows, const int32_t* weight_rows, int64_t num_edges, int64_t num_pairs,
                    int32_t width, void* edge_rows, nqa_stream stream) {
  if (num_edges < 0 || num_pairs < 0 || width <= 0 ||
      (num_edges > 0 && (!weight_rows || !edge_rows || (num_pairs > 0 && !pair_rows)))) {
    set_error("nqa_pair_expand: invalid argument");
    return NQA_ERR_INVALID;
  }
  if (num_edges == 0) return NQA_OK;
  const int64_t n = num_edges * width;
  hipLaunchKernelGGL(pair_expand_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const uint32_t*>(pair_rows), weight_rows, num_edges,
                     num_pairs, width, static_cast<uint32_t*>(edge_rows));
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_error(std::string("nqa_pair_expand: ") + hipGetErrorString(err));
    return NQA_ERR_LAUNCH;
  }
  return NQA_OK;
}

}  // extern "C"
