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
// Method: the list grouped by destination (the dst-CSR every consumer needs anyway) already holds, in row j, every edge that
// can be the reverse of an edge (i <- j, S): one thread per edge scans that row (~40 entries, contiguous) for the entries with
// source i and shift -S.  Exactly one match for EVERY edge makes the matching an involution with opposite orientations;
// representative edges are numbered in edge order by a prefix sum.  Anything else (odd E, a missing reverse edge, duplicates,
// a self edge without a shift) clears the `ok` flag and the caller keeps the per-edge evaluation.  No sort: the by-source CSR
// of a paired list is the by-destination CSR seen through the matching (nqa_csr_from_pairs).
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_scan.hpp>

#include <cstdint>
#include <string>

#include "plan.h"

namespace nqa {

static int64_t ep_align256(int64_t v) { return (v + 255) / 256 * 256; }

template <typename ST>
__device__ __forceinline__ void ep_shift(const ST* __restrict__ shift, int64_t e, int& sx, int& sy, int& sz) {
  sx = sy = sz = 0;
  if (shift != nullptr) {
    sx = (int)lrint((double)shift[3 * e + 0]);
    sy = (int)lrint((double)shift[3 * e + 1]);
    sz = (int)lrint((double)shift[3 * e + 2]);
  }
}

// partner[e] = the one edge (src[e] <- dst[e], -S); is_rep[e] = canonical orientation (dst < src; self images: first
// non-zero shift component positive).  EIGHT lanes per edge read row src[e] of the dst-CSR (~40 entries, shared through the
// cache by the ~40 edges that point into it), six entries per lane in flight at once, and compare shifts only where the source
// matches.  The kernel is a chain of dependent loads (src -> rowptr -> row -> [edge id ->] shift): its time is the number of
// wavefronts over the number the device holds, times the chain -- hence few lanes per edge and all row loads issued together.
// IDENT: the dst-CSR lists the edges in edge order (edge_id[k] == k: a list grouped by centre atom), one load less in the chain.
constexpr int kPartnerLanes = 8;
constexpr int kPartnerDepth = 6;
template <typename ST, bool IDENT>
__global__ __launch_bounds__(256) void edge_partner_kernel(const int64_t* __restrict__ dst, const int64_t* __restrict__ src,
                                                           const ST* __restrict__ shift,
                                                           const int32_t* __restrict__ rowptr,
                                                           const int32_t* __restrict__ edge_id,
                                                           const int32_t* __restrict__ src_sorted, int64_t E, int64_t N,
                                                           int32_t* __restrict__ partner, int32_t* __restrict__ is_rep,
                                                           int32_t* __restrict__ ok) {
  const int64_t e = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kPartnerLanes;
  const int sub = threadIdx.x % kPartnerLanes;
  if (e >= E) return;  // (whole groups leave together)
  const int64_t i = dst[e], j = src[e];
  int32_t match = -1, rep = 0;
  int cnt = 0;
  bool good = i >= 0 && j >= 0 && i < N && j < N;
  if (good) {
    int sx, sy, sz;
    ep_shift(shift, e, sx, sy, sz);
    if (i == j && sx == 0 && sy == 0 && sz == 0) good = false;  // a self edge without a shift has no partner
    rep = (i != j) ? (i < j ? 1 : 0) : ((sx > 0 || (sx == 0 && (sy > 0 || (sy == 0 && sz > 0)))) ? 1 : 0);
    const int32_t want = (int32_t)i;
    const int32_t k1 = rowptr[j + 1];
    for (int32_t kb = rowptr[j] + sub; kb < k1; kb += kPartnerLanes * kPartnerDepth) {
      int32_t v[kPartnerDepth];
#pragma unroll
      for (int u = 0; u < kPartnerDepth; ++u) {
        const int32_t k = kb + kPartnerLanes * u;
        v[u] = k < k1 ? src_sorted[k] : -1;
      }
#pragma unroll
      for (int u = 0; u < kPartnerDepth; ++u) {
        if (v[u] != want) continue;
        const int32_t k = kb + kPartnerLanes * u;
        const int32_t e2 = IDENT ? k : edge_id[k];
        if (e2 == (int32_t)e) continue;
        int tx, ty, tz;
        ep_shift(shift, (int64_t)e2, tx, ty, tz);
        if (tx == -sx && ty == -sy && tz == -sz) {
          match = e2;
          ++cnt;
        }
      }
    }
  }
#pragma unroll
  for (int off = kPartnerLanes / 2; off > 0; off >>= 1) {
    cnt += __shfl_xor(cnt, off, kPartnerLanes);
    const int32_t m = __shfl_xor(match, off, kPartnerLanes);
    match = m > match ? m : match;
  }
  if (sub != 0) return;
  if (cnt != 1) good = false;
  if (!good) atomicAnd(ok, 0);
  partner[e] = match >= 0 ? match : (int32_t)e;
  is_rep[e] = rep;
}

__global__ __launch_bounds__(256) void edge_pairs_number_kernel(const int32_t* __restrict__ partner,
                                                                const int32_t* __restrict__ is_rep,
                                                                const int32_t* __restrict__ pnum, int64_t E,
                                                                int32_t* __restrict__ weight_rows,
                                                                int64_t* __restrict__ rep_edge) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int32_t P = (int32_t)(E / 2);
  // (row numbers beyond the P pairs only arise on a list that does not pair up, whose arrays are not to be used: they are
  // clamped all the same, so that a caller that reads the verdict late -- a captured graph -- never holds an index out of range)
  if (is_rep[e]) {
    const int32_t p = pnum[e];
    weight_rows[e] = p < P ? p : P - 1;
    if (p < P) rep_edge[p] = e;
  } else {
    const int32_t p = pnum[partner[e]];
    weight_rows[e] = (p < P ? p : P - 1) + P;
  }
}

__global__ __launch_bounds__(256) void csr_from_pairs_kernel(const int32_t* __restrict__ edge_id_dst,
                                                             const int32_t* __restrict__ partner, int64_t E,
                                                             int32_t* __restrict__ edge_id_src) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < E) edge_id_src[k] = partner[edge_id_dst[k]];
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

// (one wavefront per node: a row is walked 64 entries at a time; a slot position is the number of qualifying entries before
// it in the row -- ballot + population count --, i.e. the order a single thread walking the row would produce)
__global__ __launch_bounds__(256) void pair_owner_count_kernel(const int32_t* __restrict__ rowptr,
                                                               const int32_t* __restrict__ edge_id,
                                                               const int32_t* __restrict__ src_sorted,
                                                               const int32_t* __restrict__ rows, int64_t N, int32_t P,
                                                               int32_t* __restrict__ cnt_own, int32_t* __restrict__ cnt_oth) {
  const int64_t n = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (n > N) return;
  const int lane = threadIdx.x & 63;
  int own = 0, oth = 0;
  if (n < N) {
    const int32_t k1 = rowptr[n + 1];
    for (int32_t kb = rowptr[n]; kb < k1; kb += 64) {
      const int32_t k = kb + lane;
      const bool valid = k < k1;
      const bool in = valid && ep_is_in((int)n, src_sorted[k], rows[edge_id[k]], P);
      own += __popcll(__builtin_amdgcn_ballot_w64(in));
      oth += __popcll(__builtin_amdgcn_ballot_w64(valid && !in));
    }
  }
  if (lane == 0) {
    cnt_own[n] = own;  // (entry N = 0: the exclusive scan over N + 1 entries ends in the total)
    cnt_oth[n] = oth;
  }
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
  const int64_t n = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (n >= N) return;
  const int lane = threadIdx.x & 63;
  int32_t base = owner_rowptr[n];
  const int32_t k1 = rowptr[n + 1];
  for (int32_t kb = rowptr[n]; kb < k1; kb += 64) {
    const int32_t k = kb + lane;
    int32_t e = 0, j = 0, r = 0;
    bool in = false;
    if (k < k1) {
      e = edge_id[k];
      j = src_sorted[k];
      r = rows[e];
      in = ep_is_in((int)n, j, r, P);
    }
    const uint64_t m = __builtin_amdgcn_ballot_w64(in);
    const int32_t s = base + __popcll(m & ((1ull << lane) - 1ull));
    if (in && s < P) {  // (s >= P: a list that does not pair up, see edge_pairs_number_kernel)
      const int32_t p = r < P ? r : r - P;
      pair_other[s] = j;
      pair_row[s] = p;
      edge_in[s] = e;
      edge_out[s] = r < P ? rev_edge[p] : (int32_t)rep_edge[p];
      slot_of_pair[p] = s;
    }
    base += __popcll(m);
  }
}

__global__ __launch_bounds__(256) void pair_other_fill_kernel(const int32_t* __restrict__ rowptr,
                                                              const int32_t* __restrict__ edge_id,
                                                              const int32_t* __restrict__ src_sorted,
                                                              const int32_t* __restrict__ rows, int64_t N, int32_t P,
                                                              const int32_t* __restrict__ other_rowptr,
                                                              const int32_t* __restrict__ slot_of_pair,
                                                              int32_t* __restrict__ other_slot) {
  const int64_t n = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (n >= N) return;
  const int lane = threadIdx.x & 63;
  int32_t base = other_rowptr[n];
  const int32_t k1 = rowptr[n + 1];
  for (int32_t kb = rowptr[n]; kb < k1; kb += 64) {
    const int32_t k = kb + lane;
    int32_t r = 0;
    bool out = false;
    if (k < k1) {
      r = rows[edge_id[k]];
      out = !ep_is_in((int)n, src_sorted[k], r, P);
    }
    const uint64_t m = __builtin_amdgcn_ballot_w64(out);
    const int32_t s = base + __popcll(m & ((1ull << lane) - 1ull));
    if (out && s < P) other_slot[s] = slot_of_pair[r < P ? r : r - P];
    base += __popcll(m);
  }
}

// Deferred verdict (a captured graph cannot wait for `ok`): a list that did not pair up gets EMPTY owner lists, so the
// pair-centric kernels walk nothing instead of following unwritten slots.  The evaluation is void either way; this keeps it
// in bounds until the caller has read the flag.
__global__ __launch_bounds__(256) void pair_lists_guard_kernel(const int32_t* __restrict__ ok, int64_t N,
                                                               int32_t* __restrict__ owner_rowptr,
                                                               int32_t* __restrict__ other_rowptr) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i > N || *ok != 0) return;
  owner_rowptr[i] = 0;
  other_rowptr[i] = 0;
}

static size_t ep_scan_bytes(int64_t N) {
  size_t bytes = 0;
  (void)rocprim::exclusive_scan(nullptr, bytes, (const int32_t*)nullptr, (int32_t*)nullptr, (int32_t)0, (size_t)(N + 1),
                                rocprim::plus<int32_t>(), (hipStream_t)0);
  return bytes;
}

static size_t ep_scan_e_bytes(int64_t E) {
  size_t bytes = 0;
  (void)rocprim::exclusive_scan(nullptr, bytes, (const int32_t*)nullptr, (int32_t*)nullptr, (int32_t)0, (size_t)E,
                                rocprim::plus<int32_t>(), (hipStream_t)0);
  return bytes;
}

}  // namespace nqa

using namespace nqa;

extern "C" {

int64_t nqa_edge_pairs_workspace_bytes(int64_t num_edges) {
  if (num_edges < 0 || num_edges > 2147483647LL) return -1;
  const int64_t E = num_edges > 0 ? num_edges : 1;
  return 2 * ep_align256(E * 4) + ep_align256((int64_t)ep_scan_e_bytes(E));
}

int nqa_edge_pairs(const int64_t* edge_dst, const int64_t* edge_src, const void* edge_cell_shift, int32_t shift_dtype,
                   const int32_t* rowptr_dst, const int32_t* edge_id_dst, const int32_t* src_sorted, int64_t num_edges,
                   int64_t num_nodes, void* workspace, int64_t workspace_bytes, int32_t* weight_rows, int64_t* rep_edge,
                   int32_t* partner_edge, int32_t* ok, nqa_stream stream) {
  if (num_edges < 0 || num_nodes < 0 || !ok ||
      (num_edges > 0 && (!edge_dst || !edge_src || !rowptr_dst || !src_sorted || !weight_rows || !rep_edge ||
                         !partner_edge)) ||
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
  int32_t* is_rep = reinterpret_cast<int32_t*>(p);
  p += ep_align256(E * 4);
  int32_t* pnum = reinterpret_cast<int32_t*>(p);
  p += ep_align256(E * 4);
  size_t scan_bytes = ep_scan_e_bytes(E);
  const unsigned ge = (unsigned)((E + 255) / 256);
  const unsigned gp = (unsigned)((E * kPartnerLanes + 255) / 256);
  const bool f32 = edge_cell_shift && shift_dtype == NQA_F32;
  const bool ident = edge_id_dst == nullptr;  // the dst-CSR lists the edges in edge order
#define NQA_PARTNER(ST, ID)                                                                                          \
  hipLaunchKernelGGL((edge_partner_kernel<ST, ID>), dim3(gp), dim3(256), 0, s, edge_dst, edge_src,                     \
                     static_cast<const ST*>(edge_cell_shift), rowptr_dst, edge_id_dst, src_sorted, E, num_nodes,      \
                     partner_edge, is_rep, ok)
  if (f32 && ident) NQA_PARTNER(float, true);
  else if (f32) NQA_PARTNER(float, false);
  else if (ident) NQA_PARTNER(double, true);
  else NQA_PARTNER(double, false);
#undef NQA_PARTNER
  if (rocprim::exclusive_scan(p, scan_bytes, is_rep, pnum, (int32_t)0, (size_t)E, rocprim::plus<int32_t>(), s) !=
      hipSuccess) {
    set_error("nqa_edge_pairs: prefix sum failed");
    return NQA_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(edge_pairs_number_kernel, dim3(ge), dim3(256), 0, s, partner_edge, is_rep, pnum, E, weight_rows,
                     rep_edge);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_error(std::string("nqa_edge_pairs: ") + hipGetErrorString(err));
    return NQA_ERR_LAUNCH;
  }
  return NQA_OK;
}

int nqa_csr_from_pairs(const int32_t* edge_id_dst, const int32_t* partner_edge, int64_t num_edges, int32_t* edge_id_src,
                       nqa_stream stream) {
  if (num_edges < 0 || (num_edges > 0 && (!edge_id_dst || !partner_edge || !edge_id_src))) {
    set_error("nqa_csr_from_pairs: invalid argument");
    return NQA_ERR_INVALID;
  }
  if (num_edges == 0) return NQA_OK;
  hipLaunchKernelGGL(csr_from_pairs_kernel, dim3((unsigned)((num_edges + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), edge_id_dst, partner_edge, num_edges, edge_id_src);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_error(std::string("nqa_csr_from_pairs: ") + hipGetErrorString(err));
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
  const unsigned gn = (unsigned)((N + 1 + 3) / 4);  // one wavefront per node
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
    hipLaunchKernelGGL(pair_owner_fill_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, rowptr_dst, edge_id_dst,
                       src_sorted, weight_rows, rep_edge, rev_edge, N, P, owner_rowptr, pair_other, pair_row, pair_edge_in,
                       pair_edge_out, slot_of_pair);
    hipLaunchKernelGGL(pair_other_fill_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, rowptr_dst, edge_id_dst,
                       src_sorted, weight_rows, N, P, other_rowptr, slot_of_pair, other_slot);
  }
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_error(std::string("nqa_pair_owner_lists: ") + hipGetErrorString(err));
    return NQA_ERR_LAUNCH;
  }
  return NQA_OK;
}

int nqa_pair_owner_lists_guard(const int32_t* ok, int64_t num_nodes, int32_t* owner_rowptr, int32_t* other_rowptr,
                               nqa_stream stream) {
  if (!ok || num_nodes < 0 || !owner_rowptr || !other_rowptr) {
    set_error("nqa_pair_owner_lists_guard: invalid argument");
    return NQA_ERR_INVALID;
  }
  hipLaunchKernelGGL(pair_lists_guard_kernel, dim3((unsigned)((num_nodes + 1 + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), ok, num_nodes, owner_rowptr, other_rowptr);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_error(std::string("nqa_pair_owner_lists_guard: ") + hipGetErrorString(err));
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
