// Node-side equivariant channel mixing and gate, fused per layer into single launches.
//
// Replaces, on the N atom rows, the e3nn modules the reference calls around the tensor product:
//   o3.Linear            linear_1 / linear_2                     (nequip/nn/interaction_block.py:82-87,129-138,177,201)
//   FullyConnectedTP     self-connection sc(x, node_attrs)      (nequip/nn/interaction_block.py:142-146,175)
//   AvgNumNeighborsNorm  x * 1/sqrt(avg_num_neighbors)          (nequip/nn/norm.py:48-68)   [folded into the weights]
//   Gate                 act(scalars) (+) act(gates) * gated    (nequip/nn/convnetlayer.py:104-112,162-164)
// which in the reference (and in a plain PyTorch port) are ~40 small ATen kernels per layer (slices, transposes,
// per-irrep GEMMs, cats, adds): 2.2 ms of the 7.8 ms cfg-3 step (profiles/).  Here every linear map of a layer is
// one launch:  out[z, ob, w, m] = scale * sum_{(ib -> ob)} sum_u x[z, ib, u, m] * W_{type(z)}[ib->ob][u, w]  (+ addend)
// in mul_ir layout.  A workgroup stages the full input rows of NZ = 8 atoms in LDS; wavefronts own 64-channel output
// chunks (lanes = output channel w), stream the weight rows coalesced from L2 and read the inputs as LDS broadcasts.
// FLOPs are tiny (~5 GFLOP per evaluation); this is a launch-count / HBM-round-trip optimisation (fp32 FMA on the
// VALU at the same rate as fp32 MFMA).  The self-connection uses per-atom-type pre-contracted weights (exact
// re-association of sum_v W[u,v,w] emb[t,v]).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "plan.h"

namespace nqa {

constexpr int kNZ = 4;  // atoms per workgroup

struct NodeChunk {  // one chunk of `width` channels (64: VALU kernel, 128: MFMA kernel) of one output irrep block
  int32_t o_off, d, mul_out, c0;
  int32_t instr_begin, instr_end, width, pad1;
};
struct NodeInstr {  // one (input block -> output block) weight matrix [mul_in, mul_out], row-major
  int32_t x_off, mul_in, w_off, pad;
};

template <typename T>
struct NodeLinearArgs {
  const T* __restrict__ x;
  const T* __restrict__ w;       // [n_types][wstride]
  const T* __restrict__ addend;  // optional [N, dout]
  T* __restrict__ out;
  const int64_t* __restrict__ types;  // optional [N]
  const NodeChunk* __restrict__ chunks;
  const NodeInstr* __restrict__ instr;
  int32_t n_chunks, n_types, din, dout;
  int64_t wstride;
  int64_t N;
  T scale;
};

// One u-block: UB consecutive input channels.  The UB weight values are loaded first (independent, coalesced L2
// reads), then the UB*D contiguous inputs of every atom are read from LDS (wide broadcast reads when aligned).
template <typename T, int D, int UB, bool VEC>
__device__ __forceinline__ void node_linear_ublock(const T* __restrict__ xu, int din, const T (&wv)[UB],
                                                   T (&acc)[kNZ][D], const int* tz, int t_sel) {
  constexpr int NV = UB * D;
#pragma unroll
  for (int z = 0; z < kNZ; ++z) {
    if (t_sel >= 0 && tz[z] != t_sel) continue;  // wave-uniform
    T xv[NV];
    if constexpr (VEC && sizeof(T) == 4) {
      const float4* __restrict__ p4 = reinterpret_cast<const float4*>(xu + z * din);
#pragma unroll
      for (int q = 0; q < NV / 4; ++q) {
        const float4 v = p4[q];
        xv[4 * q + 0] = v.x;
        xv[4 * q + 1] = v.y;
        xv[4 * q + 2] = v.z;
        xv[4 * q + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int q = 0; q < NV; ++q) xv[q] = xu[z * din + q];
    }
#pragma unroll
    for (int j = 0; j < UB; ++j)
#pragma unroll
      for (int m = 0; m < D; ++m) acc[z][m] += xv[j * D + m] * wv[j];
  }
}

template <typename T, int D>
__device__ __forceinline__ void node_linear_chunk(const NodeLinearArgs<T>& a, const NodeChunk& ch, const T* xs,
                                                  const int* tz, int64_t z0, int lane) {
  constexpr int UB = 8;
  const int wch = ch.c0 + lane;
  const bool act = wch < ch.mul_out;
  T acc[kNZ][D];
#pragma unroll
  for (int z = 0; z < kNZ; ++z)
#pragma unroll
    for (int m = 0; m < D; ++m) acc[z][m] = T(0);
  for (int q = ch.instr_begin; q < ch.instr_end; ++q) {
    const NodeInstr ins = a.instr[q];
    const T* __restrict__ wp = a.w + ins.w_off + wch;
    const bool vec = (ins.x_off % 4 == 0) && (a.din % 4 == 0) && (UB * D % 4 == 0);
    const int ufull = ins.mul_in - ins.mul_in % UB;
    for (int t = 0; t < a.n_types; ++t) {
      const T* __restrict__ wt = wp + (int64_t)t * a.wstride;
      const int t_sel = a.n_types == 1 ? -1 : t;
      // weights of the next u-block are requested before the FMAs of the current one (hides the L2 latency)
      T wn[UB];
#pragma unroll
      for (int j = 0; j < UB; ++j) wn[j] = (act && j < ufull) ? wt[(int64_t)j * ch.mul_out] : T(0);
      for (int u0 = 0; u0 < ufull; u0 += UB) {
        T wv[UB];
#pragma unroll
        for (int j = 0; j < UB; ++j) wv[j] = wn[j];
        if (u0 + UB < ufull) {
#pragma unroll
          for (int j = 0; j < UB; ++j) wn[j] = act ? wt[(int64_t)(u0 + UB + j) * ch.mul_out] : T(0);
        }
        const T* __restrict__ xu = xs + ins.x_off + u0 * D;
        if (vec) node_linear_ublock<T, D, UB, true>(xu, a.din, wv, acc, tz, t_sel);
        else node_linear_ublock<T, D, UB, false>(xu, a.din, wv, acc, tz, t_sel);
      }
      for (int u = ufull; u < ins.mul_in; ++u) {  // remainder channels
        T wv[1];
        wv[0] = act ? wt[(int64_t)u * ch.mul_out] : T(0);
        node_linear_ublock<T, D, 1, false>(xs + ins.x_off + u * D, a.din, wv, acc, tz, t_sel);
      }
    }
  }
  if (act) {
#pragma unroll
    for (int z = 0; z < kNZ; ++z) {
      if (z0 + z < a.N) {
        const int64_t o = (z0 + z) * a.dout + ch.o_off + (int64_t)wch * D;
#pragma unroll
        for (int m = 0; m < D; ++m) {
          T v = a.scale * acc[z][m];
          if (a.addend != nullptr) v += a.addend[o + m];
          a.out[o + m] = v;
        }
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void node_linear_kernel(const NodeLinearArgs<T> a) {
  extern __shared__ __align__(16) unsigned char nqa_node_smem[];
  T* xs = reinterpret_cast<T*>(nqa_node_smem);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t z0 = (int64_t)blockIdx.x * kNZ;
  const int64_t total = (int64_t)kNZ * a.din;
  const int64_t avail = (a.N - z0) * a.din;
  for (int64_t i = tid; i < total; i += 256) xs[i] = i < avail ? a.x[z0 * a.din + i] : T(0);
  int tz[kNZ];
#pragma unroll
  for (int z = 0; z < kNZ; ++z) tz[z] = (a.types != nullptr && z0 + z < a.N) ? (int)a.types[z0 + z] : 0;
  __syncthreads();
  {
    const int c = blockIdx.y * 4 + wv;  // one 64-channel output chunk per wavefront
    if (c >= a.n_chunks) return;
    const NodeChunk ch = a.chunks[c];
    switch (ch.d) {
      case 1: node_linear_chunk<T, 1>(a, ch, xs, tz, z0, lane); break;
      case 3: node_linear_chunk<T, 3>(a, ch, xs, tz, z0, lane); break;
      case 5: node_linear_chunk<T, 5>(a, ch, xs, tz, z0, lane); break;
      case 7: node_linear_chunk<T, 7>(a, ch, xs, tz, z0, lane); break;
      case 9: node_linear_chunk<T, 9>(a, ch, xs, tz, z0, lane); break;
      default: break;
    }
  }
}

// ---- float32 on the matrix cores ---------------------------------------------------------------------------------
// Transposed product  D[i = w][j = (z, m)] = sum_u A[i][u] B[u][j]  on v_mfma_f32_32x32x2_f32 (exact fp32):
//   A[i = w][k = u]   = W[u][w]               weight rows, coalesced global loads (L2 resident), one value per lane
//   B[k = u][j = z,m] = x[z, x_off + u*d + m]  node rows straight from global/L1 (each lane walks its own row)
// A wavefront owns floor(32/d) atoms (their d components fill the 32 columns) and up to four 32-channel row tiles of
// one output block, so every B value feeds up to four MFMAs.  No LDS, no barriers; masks handle ragged edges
// (odd mul, mul_out not a multiple of 32, partial atom groups) and, for the per-type self-connection, columns whose
// atom type differs from the weight set being applied.  (A register-double-buffered variant of the batch loop was
// measured slower -- 0.85 vs 0.69 ms per cfg-3 step -- and is not used.)
using f32x16n = __attribute__((ext_vector_type(16))) float;

template <int D>
__device__ __forceinline__ void node_linear_mfma_item(const NodeLinearArgs<float>& a, const NodeChunk& ch, int64_t g,
                                                      int lane) {
  constexpr int NZT = 32 / D;
  constexpr int TB = 8;  // k-pairs per register batch
  const int half = lane >> 5, j = lane & 31;
  const int zl = j / D, m = j - zl * D;
  const int64_t z = g * NZT + zl;
  const bool col_ok = (zl < NZT) && (z < a.N);
  const int tzj = (a.types != nullptr && col_ok) ? (int)a.types[z] : 0;
  const int cw = min(ch.width, ch.mul_out - ch.c0);
  const int nwt = (cw + 31) >> 5;  // 32-channel row tiles in this chunk (<= 4)
  f32x16n acc[4];
#pragma unroll
  for (int wt = 0; wt < 4; ++wt) acc[wt] = (f32x16n){0};
  for (int q = ch.instr_begin; q < ch.instr_end; ++q) {
    const NodeInstr ins = a.instr[q];
    const int K = ins.mul_in;
    const float* __restrict__ xrow = a.x + (col_ok ? z : 0) * a.din + ins.x_off + m;
    for (int t = 0; t < a.n_types; ++t) {
      const float* __restrict__ wbase = a.w + (int64_t)t * a.wstride + ins.w_off + ch.c0 + j;
      const bool bsel = col_ok && (a.n_types == 1 || tzj == t);
      for (int k0 = 0; k0 < K; k0 += 2 * TB) {
        float bq[TB], aq[TB][4];
#pragma unroll
        for (int i = 0; i < TB; ++i) {
          const int u = k0 + 2 * i + half;
          const bool uok = u < K;
          bq[i] = (bsel && uok) ? xrow[(int64_t)u * D] : 0.f;
#pragma unroll
          for (int wt = 0; wt < 4; ++wt)
            aq[i][wt] = (uok && wt < nwt && (wt * 32 + j) < cw) ? wbase[(int64_t)u * ch.mul_out + wt * 32] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < TB; ++i) {
#pragma unroll
          for (int wt = 0; wt < 4; ++wt)
            if (wt < nwt) acc[wt] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[i][wt], bq[i], acc[wt], 0, 0, 0);
        }
      }
    }
  }
  if (col_ok) {
    const int64_t obase = z * a.dout + ch.o_off + m;
#pragma unroll
    for (int wt = 0; wt < 4; ++wt) {
      if (wt >= nwt) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int wl = wt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (wl < cw) {
          const int64_t o = obase + (int64_t)(ch.c0 + wl) * D;
          float v = a.scale * acc[wt][r];
          if (a.addend != nullptr) v += a.addend[o];
          a.out[o] = v;
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void node_linear_mfma_kernel(const NodeLinearArgs<float> a) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const NodeChunk ch = a.chunks[blockIdx.y];
  const int64_t g = (int64_t)blockIdx.x * 4 + wv;
  const int nzt = 32 / ch.d;
  if (g * nzt >= a.N) return;
  switch (ch.d) {
    case 1: node_linear_mfma_item<1>(a, ch, g, lane); break;
    case 3: node_linear_mfma_item<3>(a, ch, g, lane); break;
    case 5: node_linear_mfma_item<5>(a, ch, g, lane); break;
    case 7: node_linear_mfma_item<7>(a, ch, g, lane); break;
    case 9: node_linear_mfma_item<9>(a, ch, g, lane); break;
    default: break;
  }
}

// ---- Gate ----------------------------------------------------------------------------------------------------
// in  = [scalars (ns) | gates (ng) | gated blocks (mul_b x d_b)...],  out = [act(scalars) | act(gates)[u] * gated[u, :]]
// activation id per scalar / gate segment: 0 = identity, 1 = silu * cst, 2 = tanh * cst.
template <typename T>
__device__ __forceinline__ T act_eval(int act, T x, T cst) {
  if (act == 1) return cst * x / (T(1) + exp(-x));
  if (act == 2) return cst * tanh(x);
  return x;
}
template <typename T>
__device__ __forceinline__ T act_grad(int act, T x, T cst) {
  if (act == 1) {
    const T s = T(1) / (T(1) + exp(-x));
    return cst * s * (T(1) + x * (T(1) - s));
  }
  if (act == 2) {
    const T t = tanh(x);
    return cst * (T(1) - t * t);
  }
  return T(1);
}

// Column tables (built by the host from the irreps bookkeeping, one 32-byte record per column):
//   forward,  per OUTPUT column c: {a = src, b = gate (-1: scalar), c = act, cst}
//       out[z,c] = gate < 0 ? act(in[z,src]) : act(in[z,gate]) * in[z,src]
//   backward, per INPUT column c:  {a = kind, b = act, c = o, d = i, cst, e = len, f = gate}
//       kind 0 scalar: gin = g[z,o] * act'(in[z,c])
//       kind 1 gate  : gin = act'(in[z,c]) * sum_{m<len} g[z,o+m] * in[z,i+m]
//       kind 2 gated : gin = act(in[z,gate]) * g[z,o]
// One thread per (atom, column), columns fastest: every global access is coalesced and the kernels are pure streams.
struct GateCol {
  int32_t a, b, c, d;
  double cst;  // e3nn normalize2mom constant of the activation
  int32_t e, f;
};
static_assert(sizeof(GateCol) == 32, "GateCol layout is part of the C ABI");

template <typename T>
struct GateArgs {
  const T* __restrict__ in;
  const T* __restrict__ gout;  // backward only
  T* __restrict__ out;         // fwd: out [N, dout]; bwd: gin [N, din]
  const GateCol* __restrict__ cols;
  int32_t din, dout;
  int64_t N;
};

template <typename T>
__global__ __launch_bounds__(256) void gate_fwd_kernel(const GateArgs<T> a) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.N * a.dout) return;
  const int64_t z = i / a.dout;
  const int c = (int)(i - z * a.dout);
  const GateCol t = a.cols[c];
  const T* __restrict__ row = a.in + z * a.din;
  const T cst = (T)t.cst;
  const T xs = row[t.a];
  a.out[i] = t.b < 0 ? act_eval(t.c, xs, cst) : act_eval(t.c, row[t.b], cst) * xs;
}

template <typename T>
__global__ __launch_bounds__(256) void gate_bwd_kernel(const GateArgs<T> a) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.N * a.din) return;
  const int64_t z = i / a.din;
  const int c = (int)(i - z * a.din);
  const GateCol t = a.cols[c];
  const T* __restrict__ row = a.in + z * a.din;
  const T* __restrict__ g = a.gout + z * a.dout;
  const T cst = (T)t.cst;
  if (t.a == 0) {
    a.out[i] = g[t.c] * act_grad(t.b, row[c], cst);
  } else if (t.a == 1) {
    T s = T(0);
    for (int m = 0; m < t.e; ++m) s += g[t.c + m] * row[t.d + m];
    a.out[i] = s * act_grad(t.b, row[c], cst);
  } else if (t.a == 2) {
    a.out[i] = act_eval(t.b, row[t.f], cst) * g[t.c];
  } else {
    a.out[i] = T(0);
  }
}

}  // namespace nqa

using namespace nqa;

extern "C" {

int nqa_node_linear(int32_t dtype, const void* x, const void* weights, const void* addend, void* out,
                    const int64_t* atom_types, const void* chunk_table, int32_t n_chunks, const void* instr_table,
                    int32_t n_types, int64_t weight_stride, int32_t dim_in, int32_t dim_out, int64_t num_nodes,
                    double scale, int32_t chunk_width, nqa_stream stream) {
  // chunk_width 128: float32 tables for the MFMA kernel; 64: tables for the VALU kernel (float64, or float32 by choice)
  const bool use_mfma = dtype == NQA_F32 && chunk_width == 128;
  if (chunk_width != 64 && chunk_width != 128) {
    set_error("nqa_node_linear: chunk_width must be 64 or 128");
    return NQA_ERR_INVALID;
  }
  if (dtype != NQA_F32 && dtype != NQA_F64) {
    set_error("nqa_node_linear: unsupported dtype");
    return NQA_ERR_UNSUPPORTED;
  }
  if (num_nodes < 0 || n_chunks < 0 || n_types < 1 || dim_in <= 0 || dim_out <= 0 ||
      (num_nodes > 0 && (!x || !weights || !out || !chunk_table || !instr_table)) ||
      (n_types > 1 && atom_types == nullptr)) {
    set_error("nqa_node_linear: invalid argument");
    return NQA_ERR_INVALID;
  }
  if (num_nodes == 0) return NQA_OK;
  const size_t es = dtype == NQA_F32 ? 4 : 8;
  const size_t smem = (size_t)kNZ * dim_in * es;
  if (smem > 160 * 1024 - 1024) {
    set_error("nqa_node_linear: feature rows too wide for the LDS tile");
    return NQA_ERR_UNSUPPORTED;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)((num_nodes + kNZ - 1) / kNZ), (unsigned)((n_chunks + 3) / 4));
  hipError_t err;
  if (dtype == NQA_F32) {
    NodeLinearArgs<float> a{};
    a.x = static_cast<const float*>(x);
    a.w = static_cast<const float*>(weights);
    a.addend = static_cast<const float*>(addend);
    a.out = static_cast<float*>(out);
    a.types = n_types > 1 ? atom_types : nullptr;
    a.chunks = static_cast<const NodeChunk*>(chunk_table);
    a.instr = static_cast<const NodeInstr*>(instr_table);
    a.n_chunks = n_chunks;
    a.n_types = n_types;
    a.din = dim_in;
    a.dout = dim_out;
    a.wstride = weight_stride;
    a.N = num_nodes;
    a.scale = (float)scale;
    if (use_mfma) {
      // grid.x is sized for the irrep with the fewest atoms per wavefront (d = 9 -> 3 atoms); surplus wavefronts of
      // chunks with smaller d exit immediately
      const int64_t groups = (num_nodes + 2) / 3;
      const dim3 mgrid((unsigned)((groups + 3) / 4), (unsigned)n_chunks);
      hipLaunchKernelGGL(node_linear_mfma_kernel, mgrid, dim3(256), 0, s, a);
    } else {
      if (smem > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(node_linear_kernel<float>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      hipLaunchKernelGGL(node_linear_kernel<float>, grid, dim3(256), smem, s, a);
    }
  } else {
    NodeLinearArgs<double> a{};
    a.x = static_cast<const double*>(x);
    a.w = static_cast<const double*>(weights);
    a.addend = static_cast<const double*>(addend);
    a.out = static_cast<double*>(out);
    a.types = n_types > 1 ? atom_types : nullptr;
    a.chunks = static_cast<const NodeChunk*>(chunk_table);
    a.instr = static_cast<const NodeInstr*>(instr_table);
    a.n_chunks = n_chunks;
    a.n_types = n_types;
    a.din = dim_in;
    a.dout = dim_out;
    a.wstride = weight_stride;
    a.N = num_nodes;
    a.scale = scale;
    if (smem > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(node_linear_kernel<double>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(node_linear_kernel<double>, grid, dim3(256), smem, s, a);
  }
  err = hipGetLastError();
  if (err != hipSuccess) {
    set_error(std::string("nqa_node_linear: ") + hipGetErrorString(err));
    return NQA_ERR_LAUNCH;
  }
  return NQA_OK;
}

int nqa_gate(int32_t dtype, int32_t backward, const void* input, const void* grad_out, void* out,
             const void* col_table, int32_t dim_in, int32_t dim_out, int64_t num_nodes, nqa_stream stream) {
  if (dtype != NQA_F32 && dtype != NQA_F64) {
    set_error("nqa_gate: unsupported dtype");
    return NQA_ERR_UNSUPPORTED;
  }
  if (num_nodes < 0 || (num_nodes > 0 && (!input || !out || !col_table || (backward && !grad_out))) || dim_in <= 0 ||
      dim_out <= 0) {
    set_error("nqa_gate: invalid argument");
    return NQA_ERR_INVALID;
  }
  if (num_nodes == 0) return NQA_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t total = num_nodes * (int64_t)(backward ? dim_in : dim_out);
  const unsigned grid = (unsigned)((total + 255) / 256);
#define NQA_GATE_LAUNCH(T)                                                                        \
  {                                                                                               \
    GateArgs<T> a{};                                                                              \
    a.in = static_cast<const T*>(input);                                                          \
    a.gout = static_cast<const T*>(grad_out);                                                     \
    a.out = static_cast<T*>(out);                                                                 \
    a.cols = static_cast<const GateCol*>(col_table);                                              \
    a.din = dim_in;                                                                               \
    a.dout = dim_out;                                                                             \
    a.N = num_nodes;                                                                              \
    if (backward)                                                                                 \
      hipLaunchKernelGGL(gate_bwd_kernel<T>, dim3(grid), dim3(256), 0, s, a);                     \
    else                                                                                          \
      hipLaunchKernelGGL(gate_fwd_kernel<T>, dim3(grid), dim3(256), 0, s, a);                     \
  }
  if (dtype == NQA_F32) NQA_GATE_LAUNCH(float) else NQA_GATE_LAUNCH(double)
#undef NQA_GATE_LAUNCH
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_error(std::string("nqa_gate: ") + hipGetErrorString(err));
    return NQA_ERR_LAUNCH;
  }
  return NQA_OK;
}

}  // extern "C"
