// Edge topology: stable grouping of edges by a node index (CSR), on device.
//
// Replaces the index handling the reference leaves to ATen: the row gather x[edge_src]
// (nequip/nn/_tp_scatter_base.py:36) and zeros().scatter_add_ over an expanded index
// (nequip/nn/utils.py:42-51).  Arbitrary int64 indices (unsorted, repeated, isolated nodes:
// tests/unit/nn/test_tp_scatter_kernel.py:144-149) are turned into (rowptr, edge_id, other_sorted) so that
// the tensor-product kernels can do ordered, atomics-free per-node reductions.  The sort is stable
// (ascending original edge id inside a node), which makes the floating point summation order -- and
// therefore the result -- independent of scheduling.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include <cstdint>
#include <string>

#include "plan.h"

namespace nqa {

static int64_t align256(int64_t v) { return (v + 255) / 256 * 256; }

__global__ __launch_bounds__(256) void csr_prepare_kernel(const int64_t* __restrict__ key, int64_t E, int64_t N,
                                                          int32_t* __restrict__ key32, int32_t* __restrict__ val32,
                                                          int32_t* __restrict__ status) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  int64_t k = key[e];
  if (k < 0 || k >= N) {
    if (status) atomicOr(status, 1);
    k = k < 0 ? 0 : N - 1;  // keep the sort well defined; the caller raises on the status flag
  }
  key32[e] = (int32_t)k;
  val32[e] = (int32_t)e;
}

__global__ __launch_bounds__(256) void csr_rowptr_kernel(const int32_t* __restrict__ key_sorted, int64_t E,
                                                         int64_t N, int32_t* __restrict__ rowptr) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n > N) return;
  // lower_bound(key_sorted, n)
  int64_t lo = 0, hi = E;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (key_sorted[mid] < (int32_t)n) lo = mid + 1;
    else hi = mid;
  }
  rowptr[n] = (int32_t)lo;
}

__global__ __launch_bounds__(256) void csr_gather_other_kernel(const int64_t* __restrict__ other,
                                                               const int32_t* __restrict__ edge_id, int64_t E,
                                                               int64_t N, int32_t* __restrict__ other_sorted,
                                                               int32_t* __restrict__ status) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= E) return;
  int64_t o = other[edge_id[i]];
  if (o < 0 || o >= N) {
    if (status) atomicOr(status, 2);
    o = o < 0 ? 0 : N - 1;
  }
  other_sorted[i] = (int32_t)o;
}

static int end_bit_for(int64_t N) {
  int b = 1;
  while (((int64_t)1 << b) < N && b < 31) ++b;
  return b;
}

static size_t cub_temp_bytes(int64_t E, int end_bit) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const int32_t*)nullptr, (int32_t*)nullptr, (const int32_t*)nullptr,
                                  (int32_t*)nullptr, (unsigned int)E, 0u, (unsigned int)end_bit, (hipStream_t)0);
  return bytes;
}

}  // namespace nqa

using namespace nqa;

extern "C" {

int64_t nqa_csr_workspace_bytes(int64_t num_nodes, int64_t num_edges) {
  if (num_nodes < 0 || num_edges < 0 || num_edges > 2147483647LL || num_nodes > 2147483646LL) return -1;
  if (num_edges == 0) return 256;
  const int eb = end_bit_for(num_nodes);
  return 3 * align256(num_edges * 4) + align256((int64_t)cub_temp_bytes(num_edges, eb)) + 256;
}

int nqa_csr_build(const int64_t* key, const int64_t* other, int64_t num_nodes, int64_t num_edges, int32_t* rowptr,
                  int32_t* edge_id, int32_t* other_sorted, int32_t* status_flag, void* workspace,
                  int64_t workspace_bytes, nqa_stream stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (num_nodes < 0 || num_edges < 0 || rowptr == nullptr) {
    set_error("nqa_csr_build: invalid sizes or NULL rowptr");
    return NQA_ERR_INVALID;
  }
  const int64_t need = nqa_csr_workspace_bytes(num_nodes, num_edges);
  if (need < 0) {
    set_error("nqa_csr_build: problem exceeds int32 index range");
    return NQA_ERR_UNSUPPORTED;
  }
  if (num_edges == 0) {
    hipError_t err = hipMemsetAsync(rowptr, 0, (size_t)(num_nodes + 1) * sizeof(int32_t), s);
    if (err != hipSuccess) {
      set_error(std::string("nqa_csr_build: ") + hipGetErrorString(err));
      return NQA_ERR_LAUNCH;
    }
    return NQA_OK;
  }
  if (key == nullptr || other == nullptr || edge_id == nullptr || other_sorted == nullptr) {
    set_error("nqa_csr_build: NULL operand");
    return NQA_ERR_INVALID;
  }
  if (workspace == nullptr || workspace_bytes < need) {
    set_error("nqa_csr_build: workspace missing or too small");
    return NQA_ERR_WORKSPACE;
  }
  char* ws = static_cast<char*>(workspace);
  const int64_t seg = align256(num_edges * 4);
  int32_t* key_in = reinterpret_cast<int32_t*>(ws);
  int32_t* key_out = reinterpret_cast<int32_t*>(ws + seg);
  int32_t* val_in = reinterpret_cast<int32_t*>(ws + 2 * seg);
  void* cub_tmp = ws + 3 * seg;
  const int eb = end_bit_for(num_nodes);
  size_t cub_bytes = cub_temp_bytes(num_edges, eb);

  const unsigned gridE = (unsigned)((num_edges + 255) / 256);
  hipLaunchKernelGGL(csr_prepare_kernel, dim3(gridE), dim3(256), 0, s, key, num_edges, num_nodes, key_in, val_in,
                     status_flag);
  // rocPRIM's device radix sort (stable; only the low `eb` key bits are sorted)
  hipError_t err = rocprim::radix_sort_pairs(cub_tmp, cub_bytes, (const int32_t*)key_in, key_out, (const int32_t*)val_in,
                                             edge_id, (unsigned int)num_edges, 0u, (unsigned int)eb, s);
  if (err != hipSuccess) {
    set_error(std::string("nqa_csr_build(sort): ") + hipGetErrorString(err));
    return NQA_ERR_LAUNCH;
  }
  const unsigned gridN = (unsigned)((num_nodes + 1 + 255) / 256);
  hipLaunchKernelGGL(csr_rowptr_kernel, dim3(gridN), dim3(256), 0, s, (const int32_t*)key_out, num_edges, num_nodes,
                     rowptr);
  hipLaunchKernelGGL(csr_gather_other_kernel, dim3(gridE), dim3(256), 0, s, other, (const int32_t*)edge_id,
                     num_edges, num_nodes, other_sorted, status_flag);
  err = hipGetLastError();
  if (err != hipSuccess) {
    set_error(std::string("nqa_csr_build: ") + hipGetErrorString(err));
    return NQA_ERR_LAUNCH;
  }
  return NQA_OK;
}

}  // extern "C"
