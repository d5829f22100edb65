// Node-side equivariant channel mixing and gate, fused per layer into single launches.
//
// Replaces, on the N atom rows, the e3nn modules the reference calls around the tensor product:
//   o3.Linear            linear_1 / linear_2                     (nequip/nn/interaction_block.py:82-87,129-138,177,201)
//   FullyConnectedTP     self-connection sc(x, node_attrs)      (nequip/nn/interaction_block.py:142-146,175)
//   AvgNumNeighborsNorm  x * 1/sqrt(avg_num_neighbors)          (nequip/nn/norm.py:48-68)   [the launch's `scale`]
//   Gate                 act(scalars) (+) act(gates) * gated    (nequip/nn/convnetlayer.py:104-112,162-164)
// which in the reference (and in a plain PyTorch port) are ~40 small ATen kernels per layer (slices, transposes,
// per-irrep GEMMs, cats, adds): 2.2 ms of the 7.8 ms cfg-3 step (profiles/).  Here every linear map of a layer is
// one launch:  out[z, ob, w, m] = scale * sum_{(ib -> ob)} sum_u x[z, ib, u, m] * W_{type(z)}[ib->ob][u, w]  (+ addend)
// in mul_ir layout.  float32 runs on fp32 MFMA with LDS-staged operand / result slabs (node_linear_mfma_kernel, below);
// float64 -- and float32 on request -- on a VALU kernel: a workgroup stages the full input rows of NZ = 4 atoms in LDS,
// wavefronts own 64-channel output chunks (lanes = output channel w), stream the weight rows coalesced from L2 and
// read the inputs as LDS broadcasts.  14 GFLOP per cfg-3 evaluation at 16 FLOP/byte: between the HBM and fp32-MFMA
// roofs.  The self-connection uses per-atom-type pre-contracted weights (exact re-association of
// sum_v W[u,v,w] emb[t,v]); AvgNumNeighborsNorm rides on linear_1 as the `scale` argument.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>

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

constexpr int kMaxNodeChunks = 40;  // 64-channel chunks of the output irreps per launch
constexpr int kMaxNodeInstr = 64;   // (input block -> output block) matrices per launch

template <typename T>
struct NodeLinearArgs {
  const T* __restrict__ x;
  const T* __restrict__ w;       // [n_types][wstride]
  const T* __restrict__ addend;  // optional [N, dout]
  T* __restrict__ out;
  const int64_t* __restrict__ types;  // optional [N]
  // optional [N] (per-wavefront MFMA kernels): the order in which the units walk the atoms -- row r of the launch is atom
  // perm[r].  Any permutation gives the same results; one that groups the atoms by type lets a unit skip the typed stages of
  // the types it does not hold (node_fused.h)
  const int32_t* __restrict__ perm;
  int32_t n_chunks, n_types, din, dout;
  int32_t dbg;  // ablation switches (NQA_NODE_DBG), 0 in production
  int64_t wstride;
  int64_t N;
  T scale;
  // The chunk / instruction tables travel in the kernel-argument segment (scalar loads, no dependent global round
  // trips before the first operand request).
  NodeChunk chunks[kMaxNodeChunks];
  NodeInstr instr[kMaxNodeInstr];
  int32_t blk_begin[kMaxNodeChunks + 1];  // MFMA kernel: first workgroup of every chunk (exact 1-D grid, no idle blocks)
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
// Transposed product  D[i = w][j = (z, m)] = sum_u A[i][u] B[u][j]:
//   A[i = w][k = u]   = W_t[u][c0 + w]          weight rows
//   B[k = u][j = z,m] = x[z, x_off + u*d + m]    node rows: for every atom one *contiguous* run of mul_in*d floats
// A work unit owns one 64-channel chunk of one output irrep block for floor(32/d) atoms (their d components fill the 32
// MFMA columns).  The loop runs over "stages" = (instruction, atom type, K slab); the x slab of a stage goes global ->
// registers -> LDS with a padded per-atom stride (conflict-free operand reads), the result tile goes back through the
// same slab and leaves as contiguous 64*d-float runs per atom, fused with the scale and the optional addend
// (self-connection / residual).  Ragged edges (odd mul, partial atom groups) are handled by zero-filled slabs and masked
// stores; for the per-type self-connection the columns of atoms whose type differs from the staged weight set are zeroed
// in the B operand.
// (Round 1-2 ran this as 4-wavefront workgroups with the weight slab staged in LDS and barriers around every stage --
// `node_linear_mfma_kernel`, 225 VGPRs, 2 wavefronts per SIMD; the per-wavefront kernels below replaced it in round 3:
// same time in exact fp32, 0-18 % faster with split-bf16 operands depending on the box)
using f32x16n = __attribute__((ext_vector_type(16))) float;

constexpr int kNLW = 64;                       // output channels per chunk (two 32-row MFMA tiles)
constexpr int kNLXS = 32 * (kNLW + 1);         // floats per wavefront slab: NZT atoms x (64+1)*d, NZT*d <= 32

// ---- float32 on the matrix cores, one independent pipeline per wavefront (v2) ---------------------------------------
// Same product as node_linear_mfma_kernel -- D[i = w][j = (z, m)] = sum_u W_t[u][c0 + w] x[z, x_off + u d + m] on
// v_mfma_f32_32x32x2_f32, bitwise the same fma chain -- organised for the small N of a single box (10^4 atoms = a few
// thousand tiles) instead of for a large GEMM:
//   * a work unit = (64-channel output chunk, floor(32/d) atoms) belongs to ONE wavefront (64-thread workgroups): no
//     workgroup barrier anywhere, the hardware dispatcher balances the units over the SIMDs, and with <= 128 VGPRs and
//     8.2 KiB of LDS per wavefront four of them share a SIMD -- while one waits for its operands the others own the
//     matrix pipe (the v1 kernel: 2 wavefronts per SIMD, 225 VGPRs, 65 KiB LDS per 4-wavefront workgroup, barriers
//     around every stage, MFMA pipe 30 % busy);
//   * the weight fragment A[i = w][k = u] = W[u][c0 + w] is read straight from global memory / L2 in MFMA layout: the 32
//     lanes of a half-wave read 32 consecutive floats of one weight row -- full 128-byte requests without any staging;
//     batches of 8 k-steps are requested two batches ahead of the MFMAs that consume them;
//   * the x slab (for every atom one contiguous run of 64 d floats) goes global -> registers -> the wavefront's private
//     LDS slab (padded stride: conflict-free B-operand reads) and is requested one stage ahead; LDS operations of one
//     wavefront complete in order, so no waits besides the data dependencies;
//   * the result tile leaves through the same slab as contiguous runs, fused with scale and addend.
// Units are enumerated chunk by chunk, chunks with the most stages first (the long units start first).
constexpr int kNLK2 = 32;                      // input channels per stage of the per-wavefront kernel

template <int D>
__device__ __forceinline__ void node_linear_wave_unit(const NodeLinearArgs<float>& a, const NodeChunk& ch, int64_t g,
                                                      float* __restrict__ xs) {
  constexpr int NZT = 32 / D;                               // atoms per unit
  constexpr int P = D == 1 ? 1 : ((D + 3) / 4) * 4;
  constexpr int S = kNLK2 * D + P;                           // padded slab stride per atom (see v1)
  constexpr bool kVecLds = (S % 4) == 0;
  constexpr int RUN4 = kNLK2 * D / 4;                       // float4 per atom run (32*d floats)
  constexpr int XV4 = (NZT * RUN4 + 63) / 64;               // float4 per lane per slab
  constexpr int KB = 4;                                     // k-steps per weight batch
  constexpr int NB = kNLK2 / 2 / KB;                         // batches per stage (4)
  static_assert(NZT * S <= kNLXS, "slab too small");
  const int lane = threadIdx.x & 63;
  const int half = lane >> 5, j = lane & 31;
  const int zlr = j / D, m = j - zlr * D;
  const int zl = min(zlr, NZT - 1);
  const int cw = min(kNLW, ch.mul_out - ch.c0);
  const int cj0 = min(j, cw - 1), cj1 = min(j + 32, cw - 1);  // clamped weight columns (rows >= cw are never stored)
  const int64_t zbase = g * NZT;
  auto atom = [&](int64_t r) -> int64_t {  // row of the launch -> atom (clamped rows: loads from valid addresses)
    const int64_t rc = r < a.N ? r : a.N - 1;
    return a.perm != nullptr ? (int64_t)a.perm[rc] : rc;
  };
  const bool col_ok = (zlr < NZT) && (zbase + zl < a.N);
  const int64_t z = atom(zbase + zl);
  const int tzj = (a.types != nullptr && col_ok) ? (int)a.types[z] : 0;
  // atom types this unit holds (bit t; types beyond 31 always count as present): stages of absent types are skipped
  unsigned present = 0xffffffffu;
  if (a.types != nullptr && a.n_types > 1 && a.n_types <= 32) {
    present = 0;
    for (int tt = 0; tt < a.n_types; ++tt) present |= (__any(col_ok && tzj == tt) ? 1u : 0u) << tt;
  }

  int xz[XV4], xo[XV4], zrow[XV4];
#pragma unroll
  for (int v = 0; v < XV4; ++v) {
    const int idx = lane + v * 64;
    xz[v] = idx / RUN4;
    xo[v] = (idx - xz[v] * RUN4) * 4;
    zrow[v] = (int)atom(zbase + min(xz[v], NZT - 1));  // (looked up once per unit, not per stage)
  }
  auto slot_ok = [&](int v) { return (v + 1) * 64 <= NZT * RUN4 || xz[v] < NZT; };

  // stage enumeration: (instruction q, type t, K slab k0)
  int q = ch.instr_begin, t = 0, k0 = 0;
  if (q >= ch.instr_end) {  // no instruction feeds this block: zeros (+ addend)
    // falls through to the epilogue with zero accumulators
  }
  NodeInstr ins = q < ch.instr_end ? a.instr[q] : NodeInstr{0, 0, 0, 0};
  auto settle = [&]() {  // (k0 == 0) on to the next (instruction, type) whose type some atom of this unit has
    while (q < ch.instr_end && a.n_types > 1) {
      while (t < a.n_types && t < 32 && ((present >> t) & 1u) == 0u) ++t;
      if (t < a.n_types) break;
      t = 0;
      if (++q < ch.instr_end) ins = a.instr[q];
    }
  };
  settle();
  auto advance = [&]() {
    k0 += kNLK2;
    if (k0 >= ins.mul_in) {
      k0 = 0;
      if (++t >= a.n_types) {
        t = 0;
        if (++q < ch.instr_end) ins = a.instr[q];
      }
      settle();
    }
    return q < ch.instr_end;
  };

  // x slab of a stage: XV4 UNCONDITIONAL loads per lane from clamped (always valid) addresses -- no branch, no
  // per-element predicate, so the requests of the next stage really stay in flight behind the MFMAs of the current one
  // (with predicated loads the compiler merged the paths through register copies and waited for every load right
  // where it was issued).  What lies outside the slab (atoms beyond N, channels beyond mul_in) is zeroed when the
  // registers are written to LDS.
  float4 xreg[XV4];
  int xkk = 0;  // valid floats per atom of the slab held in xreg
  const bool xal_all = (a.din & 3) == 0;
  auto load_x = [&](const NodeInstr& si, int sk0) {
    const int kk = min(kNLK2, si.mul_in - sk0) * D;
    xkk = kk;
    const float* __restrict__ xb0 = a.x + si.x_off + sk0 * D;
    if (xal_all && ((si.x_off | kk) & 3) == 0) {  // wave-uniform: aligned runs, whole float4s
#pragma unroll
      for (int v = 0; v < XV4; ++v) {
        const int64_t zg = zrow[v];
        const int eo = min(xo[v], kk - 4);
        xreg[v] = *reinterpret_cast<const float4*>(xb0 + zg * a.din + eo);
      }
    } else {  // odd multiplicities / unaligned blocks: dword loads, each clamped on its own
#pragma unroll
      for (int v = 0; v < XV4; ++v) {
        const int64_t zg = zrow[v];
        const float* __restrict__ p = xb0 + zg * a.din;
        xreg[v].x = p[min(xo[v] + 0, kk - 1)];
        xreg[v].y = p[min(xo[v] + 1, kk - 1)];
        xreg[v].z = p[min(xo[v] + 2, kk - 1)];
        xreg[v].w = p[min(xo[v] + 3, kk - 1)];
      }
    }
  };
  auto store_x = [&]() {
#pragma unroll
    for (int v = 0; v < XV4; ++v) {
      if (slot_ok(v)) {
        const bool zok = xz[v] < NZT && zbase + xz[v] < a.N;
        float4 r;
        r.x = (zok && xo[v] + 0 < xkk) ? xreg[v].x : 0.f;
        r.y = (zok && xo[v] + 1 < xkk) ? xreg[v].y : 0.f;
        r.z = (zok && xo[v] + 2 < xkk) ? xreg[v].z : 0.f;
        r.w = (zok && xo[v] + 3 < xkk) ? xreg[v].w : 0.f;
        float* __restrict__ d = xs + xz[v] * S + xo[v];
        if constexpr (kVecLds) {
          *reinterpret_cast<float4*>(d) = r;
        } else {
          d[0] = r.x; d[1] = r.y; d[2] = r.z; d[3] = r.w;
        }
      }
    }
  };

  f32x16n acc0 = {0}, acc1 = {0};
  bool have = q < ch.instr_end;
  if (have) load_x(ins, k0);
  while (have) {
    // ---- this stage: (ins, t, k0) -> cur; then look ahead
    const NodeInstr cur = ins;
    const int cur_t = t, cur_k0 = k0;
    store_x();  // (waits for the slab's global loads; earlier B reads of this slab were issued before: in-order LDS)
    have = advance();
    const float* __restrict__ wb = a.w + (int64_t)cur_t * a.wstride + cur.w_off + ch.c0;
    const int ulast = cur.mul_in - 1;
    float a0[2][KB], a1[2][KB];
    auto load_w = [&](int b, int buf) {
#pragma unroll
      for (int i = 0; i < KB; ++i) {
        const int u = min(cur_k0 + 2 * (b * KB + i) + half, ulast);  // (rows beyond mul_in meet zero-filled x)
        const float* __restrict__ wr = wb + (int64_t)u * ch.mul_out;
        a0[buf][i] = wr[cj0];
        a1[buf][i] = wr[cj1];
      }
    };
    load_w(0, 0);
    load_w(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (have) load_x(ins, k0);  // next stage's slab: lands behind this stage's MFMAs
    __builtin_amdgcn_sched_barrier(0);
    const bool bsel = col_ok && (a.n_types == 1 || tzj == cur_t);
    const float* __restrict__ xb = xs + zl * S + m;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      float bq[KB];
#pragma unroll
      for (int i = 0; i < KB; ++i) bq[i] = xb[(2 * (b * KB + i) + half) * D];
#pragma unroll
      for (int i = 0; i < KB; ++i) {
        const float bv = bsel ? bq[i] : 0.f;
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[b & 1][i], bv, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[b & 1][i], bv, acc1, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (b + 2 < NB) load_w(b + 2, b & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- epilogue: result tile -> slab as [atom][w*d + m] (64 channels: stride SE), then contiguous runs per atom
  constexpr int SE = kNLW * D + P;
  constexpr bool kVecE = (SE % 4) == 0;
  constexpr int RUN4E = kNLW * D / 4;
  constexpr int XV4E = (NZT * RUN4E + 63) / 64;
  static_assert(NZT * SE <= kNLXS, "result slab too small");
  if (zlr < NZT) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int wl = (r & 3) + 8 * (r >> 2) + 4 * half;
      xs[zl * SE + wl * D + m] = acc0[r];
      xs[zl * SE + (wl + 32) * D + m] = acc1[r];
    }
  }
  const int run = cw * D;
  const bool oal = ((a.dout | ch.o_off | (ch.c0 * D)) & 3) == 0;
  const bool fast = oal && cw == kNLW && zbase + NZT <= a.N && kVecE;
#pragma unroll
  for (int v = 0; v < XV4E; ++v) {
    const int idx = lane + v * 64;
    const int ez = idx / RUN4E;
    const int eo = (idx - ez * RUN4E) * 4;
    const int64_t zr = zbase + ez;  // row of the launch
    if (ez >= NZT) continue;
    const int64_t zg = atom(zr);
    const float* __restrict__ sp = xs + ez * SE + eo;
    const int64_t o = zg * a.dout + ch.o_off + (int64_t)ch.c0 * D + eo;
    if (fast) {
      float4 r = *reinterpret_cast<const float4*>(sp);
      r.x *= a.scale; r.y *= a.scale; r.z *= a.scale; r.w *= a.scale;
      if (a.addend != nullptr) {
        const float4 ad = *reinterpret_cast<const float4*>(a.addend + o);
        r.x += ad.x; r.y += ad.y; r.z += ad.z; r.w += ad.w;
      }
      *reinterpret_cast<float4*>(a.out + o) = r;
    } else if (zr < a.N && eo < run) {
      float4 r;
      if constexpr (kVecE) {
        r = *reinterpret_cast<const float4*>(sp);
      } else {
        r = make_float4(sp[0], sp[1], sp[2], sp[3]);
      }
      r.x *= a.scale; r.y *= a.scale; r.z *= a.scale; r.w *= a.scale;
      if (oal && eo + 3 < run) {
        if (a.addend != nullptr) {
          const float4 ad = *reinterpret_cast<const float4*>(a.addend + o);
          r.x += ad.x; r.y += ad.y; r.z += ad.z; r.w += ad.w;
        }
        *reinterpret_cast<float4*>(a.out + o) = r;
      } else {
        const float rv[4] = {r.x, r.y, r.z, r.w};
        for (int e = 0; e < 4; ++e)
          if (eo + e < run) a.out[o + e] = rv[e] + (a.addend != nullptr ? a.addend[o + e] : 0.f);
      }
    }
  }
}

__global__ __launch_bounds__(64, 4) void node_linear_wave_kernel(const NodeLinearArgs<float> a) {
  __shared__ __align__(16) float xs[kNLXS];  // staging slab (32-channel K slabs) and, at the end, the 64-channel result tile
  // exact 1-D grid: chunk c owns the units [blk_begin[c], blk_begin[c+1]) (one unit = one wavefront = one workgroup)
  int c = 0;
  while (c + 1 < a.n_chunks && (int)blockIdx.x >= a.blk_begin[c + 1]) ++c;
  const NodeChunk ch = a.chunks[c];
  const int64_t g = (int64_t)((int)blockIdx.x - a.blk_begin[c]);
  switch (ch.d) {
    case 1: node_linear_wave_unit<1>(a, ch, g, xs); break;
    case 3: node_linear_wave_unit<3>(a, ch, g, xs); break;
    case 5: node_linear_wave_unit<5>(a, ch, g, xs); break;
    case 7: node_linear_wave_unit<7>(a, ch, g, xs); break;
    case 9: node_linear_wave_unit<9>(a, ch, g, xs); break;
    default: break;
  }
}

// ---- float32 accuracy on the bf16 matrix pipe (split operands), per-wavefront pipeline (v3) -------------------------
// The fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at the vector rate -- 64 FLOP/clk/SIMD, ~125 TFLOP/s at the clock the chip
// sustains with real operands -- and the per-wavefront kernel above is bound by exactly that pipe (timeline: 75-80 % busy
// in steady state; 3.4 GFLOP of the cfg-3 linear_2 cannot take less than ~30 us, the 12672 -> 2432 map of the l_max = 3
// model not less than 0.9 ms per 20 000 atoms).  As in the radial MLP (radial_mlp.hip, second half) every fp32 operand is
// written as the exact sum of three bf16 numbers and a product is accumulated in fp32 from the six partial products of
// weight >= 2^-16 on v_mfma_f32_32x32x16_bf16: 6 instructions x 32 cycles per 32x32x16 block instead of 8 x 64.
//   * weights: split and laid out in A-fragment order ONCE per weight version by nqa_node_weights_pack
//     (Wf[type][instr][k16][col tile][plane][lane] = 8 bf16 of W[16 k16 + 8 (lane >> 5) + e][32 tile + (lane & 31)]):
//     one 1 KiB wave read per fragment, straight from L2 into the MFMA operand registers;
//   * x: staged through the wavefront's LDS slab as before; the 8 k-values of a lane are read from the slab and split
//     in registers (11 VALU per pair of values);
//   * the rest (units, enumeration, epilogue) is the per-wavefront kernel above.
__device__ __forceinline__ uint32_t nl_cvt_pk_bf16(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));  // round-to-nearest-even, lo -> bits [15:0]
  return r;
}
__device__ __forceinline__ void nl_split_pair(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = nl_cvt_pk_bf16(x0, x1);
  float r0 = x0 - __uint_as_float(h << 16);
  float r1 = x1 - __uint_as_float(h & 0xffff0000u);
  m = nl_cvt_pk_bf16(r0, r1);
  r0 -= __uint_as_float(m << 16);
  r1 -= __uint_as_float(m & 0xffff0000u);
  l = nl_cvt_pk_bf16(r0, r1);
}
using nl_bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using nl_u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
__device__ __forceinline__ f32x16n nl_mfma_bf16(const nl_u32x4& a, const nl_u32x4& b, const f32x16n& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(nl_bf16x8, a), __builtin_bit_cast(nl_bf16x8, b), c,
                                                 0, 0, 0);
}

// ---- two-plane fp16 split with a running per-column scale (default; NQA_NODE_F16=0 keeps the bf16 split) ----------------
// As in the radial MLP's backward (radial_mlp.hip): x = h + l with h = fp16(x), l = fp16(x - h) carries 22 significand
// bits, three products h.h + h.l + l.h per fp32 product instead of six.  fp16's range is bridged by exact powers of two:
// every 16-row K block of a weight matrix is stored multiplied by 2^e (largest magnitude into [2^14, 2^15); e per (atom
// type, instruction, K block), kept behind the fragments), and every output COLUMN (atom, component) -- a lane of the
// accumulator layout, so all of this is lane-local -- carries a running exponent S: its accumulators hold 2^S x the true
// sums, the 16 x-values of a K block are multiplied by 2^(S - e) before they are split, and when a block's largest
// magnitude would leave the range S drops (three bits below the limit) and the lane's accumulators are multiplied by the
// bridging power of two.  The epilogue multiplies by 2^-S.
using nl_f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using nl_f16x2 = __attribute__((ext_vector_type(2))) _Float16;
__device__ __forceinline__ f32x16n nl_mfma_f16(const nl_u32x4& a, const nl_u32x4& b, const f32x16n& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(nl_f16x8, a), __builtin_bit_cast(nl_f16x8, b), c, 0,
                                                0, 0);
}
__device__ __forceinline__ void nl_split_pair_f16(float x0, float x1, uint32_t& h, uint32_t& l) {
  const nl_f16x2 hh = {(_Float16)x0, (_Float16)x1};
  const nl_f16x2 ll = {(_Float16)(x0 - (float)hh[0]), (_Float16)(x1 - (float)hh[1])};
  h = __builtin_bit_cast(uint32_t, hh);
  l = __builtin_bit_cast(uint32_t, ll);
}

struct NodePackArgs {
  const float* __restrict__ w;  // [n_types][wstride]
  nl_u32x4* __restrict__ out;   // [n_types][frag_stride]
  int32_t* __restrict__ wexp;   // F16: [n_types][exp_stride] exponent of every (instruction, 16-row K block)
  int64_t wstride, frag_stride;
  int32_t n_instr, n_types, exp_stride;
  int32_t exp_off[kMaxNodeInstr];
  int32_t mul_in[kMaxNodeInstr], mul_out[kMaxNodeInstr], w_off[kMaxNodeInstr];
  int32_t frag_off[kMaxNodeInstr + 1];  // first fragment-lane (uint4 index) of every instruction, within one type
};

// F16: exponent of every (type, instruction, K block): one wavefront each, over the block's 16 x mul_out values (rows of
// mul_out contiguous floats: coalesced)
__global__ __launch_bounds__(64) void node_weights_exp_kernel(const NodePackArgs a) {
  const int idx = (int)blockIdx.x;  // (type, exponent slot)
  const int lane = (int)threadIdx.x;
  const int t = idx / a.exp_stride, r = idx - t * a.exp_stride;
  int q = 0;
  while (q + 1 < a.n_instr && r >= a.exp_off[q + 1]) ++q;
  const int k16 = r - a.exp_off[q];
  const float* __restrict__ wq = a.w + (int64_t)t * a.wstride + a.w_off[q];
  const int u1 = min(16 * k16 + 16, a.mul_in[q]);
  float m = 0.f;
  for (int u = 16 * k16; u < u1; ++u)
    for (int c = lane; c < a.mul_out[q]; c += 64) m = fmaxf(m, fabsf(wq[(int64_t)u * a.mul_out[q] + c]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if (lane != 0) return;
  int ex = 0;
  if (m > 0.f && m < 3.0e38f) {
    int e;
    (void)frexpf(m, &e);
    ex = 15 - e;
    ex = ex > 100 ? 100 : (ex < -100 ? -100 : ex);
  }
  a.wexp[idx] = ex;
}

__global__ __launch_bounds__(256) void node_weights_pack_f16_kernel(const NodePackArgs a) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (type, fragment (q, k16, tile), lane)
  const int64_t per_type = a.frag_stride / 2;                    // one thread writes the two planes
  if (idx >= per_type * a.n_types) return;
  const int t = (int)(idx / per_type);
  const int64_t r = idx - (int64_t)t * per_type;
  int q = 0;
  while (q + 1 < a.n_instr && r * 2 >= a.frag_off[q + 1]) ++q;
  const int64_t f = r - a.frag_off[q] / 2;  // (k16 * nct + tile) * 64 + lane
  const int lane = (int)(f & 63);
  const int nct = (a.mul_out[q] + 31) / 32;
  const int tile = (int)((f >> 6) % nct), k16 = (int)((f >> 6) / nct);
  const int c = 32 * tile + (lane & 31);
  const int u0 = 16 * k16 + 8 * (lane >> 5);
  const float* __restrict__ wq = a.w + (int64_t)t * a.wstride + a.w_off[q];
  const float su = ldexpf(1.f, a.wexp[t * a.exp_stride + a.exp_off[q] + k16]);
  nl_u32x4 h, l;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int u = u0 + 2 * e;
    const float v0 = (c < a.mul_out[q] && u < a.mul_in[q]) ? wq[(int64_t)u * a.mul_out[q] + c] * su : 0.f;
    const float v1 = (c < a.mul_out[q] && u + 1 < a.mul_in[q]) ? wq[(int64_t)(u + 1) * a.mul_out[q] + c] * su : 0.f;
    uint32_t x, y;
    nl_split_pair_f16(v0, v1, x, y);
    h[e] = x; l[e] = y;
  }
  nl_u32x4* __restrict__ o = a.out + (int64_t)t * a.frag_stride + a.frag_off[q] + ((f >> 6) * 2) * 64 + lane;
  o[0] = h; o[64] = l;
}

__global__ __launch_bounds__(256) void node_weights_pack_kernel(const NodePackArgs a) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (type, fragment (q, k16, tile), lane)
  const int64_t per_type = a.frag_stride / 3;                    // one thread writes the three planes
  if (idx >= per_type * a.n_types) return;
  const int t = (int)(idx / per_type);
  const int64_t r = idx - (int64_t)t * per_type;
  int q = 0;
  while (q + 1 < a.n_instr && r * 3 >= a.frag_off[q + 1]) ++q;
  const int64_t f = r - a.frag_off[q] / 3;  // (k16 * nct + tile) * 64 + lane
  const int lane = (int)(f & 63);
  const int nct = (a.mul_out[q] + 31) / 32;
  const int tile = (int)((f >> 6) % nct), k16 = (int)((f >> 6) / nct);
  const int c = 32 * tile + (lane & 31);
  const int u0 = 16 * k16 + 8 * (lane >> 5);
  const float* __restrict__ wq = a.w + (int64_t)t * a.wstride + a.w_off[q];
  nl_u32x4 h, m, l;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int u = u0 + 2 * e;
    const float v0 = (c < a.mul_out[q] && u < a.mul_in[q]) ? wq[(int64_t)u * a.mul_out[q] + c] : 0.f;
    const float v1 = (c < a.mul_out[q] && u + 1 < a.mul_in[q]) ? wq[(int64_t)(u + 1) * a.mul_out[q] + c] : 0.f;
    uint32_t x, y, z;
    nl_split_pair(v0, v1, x, y, z);
    h[e] = x; m[e] = y; l[e] = z;
  }
  nl_u32x4* __restrict__ o = a.out + (int64_t)t * a.frag_stride + a.frag_off[q] + ((f >> 6) * 3) * 64 + lane;
  o[0] = h; o[64] = m; o[128] = l;
}

struct NodeLinearPackedArgs {
  NodeLinearArgs<float> base;            // base.w is unused
  const nl_u32x4* __restrict__ wf;       // packed weights
  const int32_t* __restrict__ wexp;      // F16: [n_types][exp_stride], index base.instr[q].pad + K block
  int32_t exp_stride;
  int64_t frag_stride;                   // uint4 per atom type
  int32_t frag_off[kMaxNodeInstr];       // uint4 offset of every instruction's fragments
  // unit enumeration: a group = the 64-channel chunks of ONE output block (same instructions, same x columns); unit u of a
  // group is (atom group u / n, chunk u % n) -- the chunks that read the same x slab run next to each other (same
  // workgroup / neighbouring workgroups), so the slab comes out of L1 / L2 for all but the first of them
  int32_t n_groups;
  int32_t grp_begin[kMaxNodeChunks + 1];  // first unit of every group
  int32_t grp_chunk0[kMaxNodeChunks];     // first chunk of the group in base.chunks
  int32_t grp_n[kMaxNodeChunks];          // chunks in the group
};

constexpr int kNLK3 = 32;  // input channels per stage of the split-bf16 kernel (64-channel slabs were measured: 148 spilled registers, slower)

struct NodeStage {  // one (instruction, atom type, 32-channel K slab) step of a chunk
  int q, t, k0, mul_in, x_off;
  bool valid;
};

// PIPE (F16 only; round 4): the stage loop restructured around how the memory counter retires.  vmcnt retires IN ORDER, and
// across the branches of this loop the compiler waits with vmcnt(0): in the loop above the weight fragments of a stage's
// second K block are requested AFTER the next stage's x slab, so waiting for them drains the slab prefetch as well -- every
// stage pays the full HBM latency twice (ISA: `s_waitcnt vmcnt(0)` in front of both MFMA groups; per-unit timeline 8-10 k
// cycles per stage for ~1.5 k cycles of issue).  Here a stage has ONE wait, at its top, for loads that were all requested a
// full stage earlier: x slab AND both K blocks' fragments of stage s+1 go out right after the slab of stage s has been
// written to LDS, into a second fragment buffer (the stage loop is unrolled by two so that both buffers are addressed
// statically); inside a stage nothing waits on memory.  64 fragment registers instead of 16: two wavefronts per SIMD.
template <int D, bool F16, bool PIPE = false>
__device__ __forceinline__ void node_linear_wave_bf16_unit(const NodeLinearPackedArgs& pa, const NodeChunk& ch, int64_t g,
                                                           float* __restrict__ xs, int unit) {
  constexpr int NPL = F16 ? 2 : 3;  // operand planes
  const NodeLinearArgs<float>& a = pa.base;
  constexpr int NZT = 32 / D;
  constexpr int P = D == 1 ? 1 : ((D + 3) / 4) * 4;
  constexpr int S = kNLK3 * D + P;
  constexpr bool kVecLds = (S % 4) == 0;
  constexpr int RUN4 = kNLK3 * D / 4;
  constexpr int XV4 = (NZT * RUN4 + 63) / 64;
  static_assert(NZT * S <= kNLXS, "slab too small");
  const int lane = threadIdx.x & 63;
  const int half = lane >> 5, j = lane & 31;
  const int zlr = j / D, m = j - zlr * D;
  const int zl = min(zlr, NZT - 1);
  const int cw = min(kNLW, ch.mul_out - ch.c0);
  const int64_t zbase = g * NZT;
  auto atom = [&](int64_t r) -> int64_t {  // row of the launch -> atom (clamped rows: loads from valid addresses)
    const int64_t rc = r < a.N ? r : a.N - 1;
    return a.perm != nullptr ? (int64_t)a.perm[rc] : rc;
  };
  const bool col_ok = (zlr < NZT) && (zbase + zl < a.N);
  const int64_t z = atom(zbase + zl);
  const int tzj = (a.types != nullptr && col_ok) ? (int)a.types[z] : 0;
  unsigned present = 0xffffffffu;  // atom types this unit holds: stages of absent types are skipped
  if (a.types != nullptr && a.n_types > 1 && a.n_types <= 32) {
    present = 0;
    for (int tt = 0; tt < a.n_types; ++tt) present |= (__any(col_ok && tzj == tt) ? 1u : 0u) << tt;
  }
  const int nct = (ch.mul_out + 31) / 32;   // column tiles of the whole output block
  const int ct0 = ch.c0 / 32;               // first tile of this chunk (c0 is a multiple of 64)
  const bool two_tiles = cw > 32;           // wave-uniform

  int xz[XV4], xo[XV4], zrow[XV4];
#pragma unroll
  for (int v = 0; v < XV4; ++v) {
    const int idx = lane + v * 64;
    xz[v] = idx / RUN4;
    xo[v] = (idx - xz[v] * RUN4) * 4;
    zrow[v] = (int)atom(zbase + min(xz[v], NZT - 1));  // (looked up once per unit, not per stage)
  }
  auto slot_ok = [&](int v) { return (v + 1) * 64 <= NZT * RUN4 || xz[v] < NZT; };

  // stage enumeration
  auto settle = [&](NodeStage st) {  // (st.k0 == 0) on to the next (instruction, type) whose type this unit holds
    while (st.valid && a.n_types > 1) {
      while (st.t < a.n_types && st.t < 32 && ((present >> st.t) & 1u) == 0u) ++st.t;
      if (st.t < a.n_types) break;
      st.t = 0;
      if (++st.q < ch.instr_end) {
        st.mul_in = a.instr[st.q].mul_in;
        st.x_off = a.instr[st.q].x_off;
      } else {
        st.valid = false;
      }
    }
    return st;
  };
  auto first_stage = [&]() {
    NodeStage st{ch.instr_begin, 0, 0, 0, 0, ch.instr_begin < ch.instr_end};
    if (st.valid) {
      st.mul_in = a.instr[st.q].mul_in;
      st.x_off = a.instr[st.q].x_off;
      st = settle(st);
    }
    return st;
  };
  auto next_stage = [&](NodeStage st) {
    if (!st.valid) return st;
    st.k0 += kNLK3;
    if (st.k0 >= st.mul_in) {
      st.k0 = 0;
      if (++st.t >= a.n_types) {
        st.t = 0;
        if (++st.q < ch.instr_end) {
          st.mul_in = a.instr[st.q].mul_in;
          st.x_off = a.instr[st.q].x_off;
        } else {
          st.valid = false;
        }
      }
      st = settle(st);
    }
    return st;
  };

  // x slab of a stage: XV4 UNCONDITIONAL loads per lane from clamped (always valid) addresses; what lies outside the
  // slab is zeroed when the registers go to LDS.  TWO slabs are kept in flight per wavefront (the stage being computed is
  // in LDS, the next two are on their way): the loaded latency of these 128 d-byte row pieces is 2-6 us (timeline:
  // first slab in LDS 7 k cycles after the unit starts), and with one 4 KiB slab in flight per wavefront the whole
  // kernel ran at latency x occupancy = 2.4 TB/s.
  const bool xal_all = (a.din & 3) == 0;
  auto load_x = [&](float4 (&xr)[XV4], const NodeStage& st) {
    const int kk = min(kNLK3, st.mul_in - st.k0) * D;
    const float* __restrict__ xb0 = a.x + st.x_off + st.k0 * D;
    if (xal_all && ((st.x_off | kk) & 3) == 0) {
#pragma unroll
      for (int v = 0; v < XV4; ++v) {
        const int64_t zg = zrow[v];
        const int eo = min(xo[v], kk - 4);
        xr[v] = *reinterpret_cast<const float4*>(xb0 + zg * a.din + eo);
      }
    } else {
#pragma unroll
      for (int v = 0; v < XV4; ++v) {
        const int64_t zg = zrow[v];
        const float* __restrict__ p = xb0 + zg * a.din;
        xr[v].x = p[min(xo[v] + 0, kk - 1)];
        xr[v].y = p[min(xo[v] + 1, kk - 1)];
        xr[v].z = p[min(xo[v] + 2, kk - 1)];
        xr[v].w = p[min(xo[v] + 3, kk - 1)];
      }
    }
  };
  auto store_x = [&](const float4 (&xr)[XV4], const NodeStage& st) {
    const int xkk = min(kNLK3, st.mul_in - st.k0) * D;
    // (wave-uniform) a complete slab of a complete atom group needs no masking
    const bool whole = xkk == kNLK3 * D && zbase + NZT <= a.N;
#pragma unroll
    for (int v = 0; v < XV4; ++v) {
      if (slot_ok(v)) {
        float4 r = xr[v];
        if (!whole) {
          const bool zok = xz[v] < NZT && zbase + xz[v] < a.N;
          r.x = (zok && xo[v] + 0 < xkk) ? r.x : 0.f;
          r.y = (zok && xo[v] + 1 < xkk) ? r.y : 0.f;
          r.z = (zok && xo[v] + 2 < xkk) ? r.z : 0.f;
          r.w = (zok && xo[v] + 3 < xkk) ? r.w : 0.f;
        }
        float* __restrict__ d = xs + xz[v] * S + xo[v];
        if constexpr (kVecLds) {
          *reinterpret_cast<float4*>(d) = r;
        } else {
          d[0] = r.x; d[1] = r.y; d[2] = r.z; d[3] = r.w;
        }
      }
    }
  };

  // A fragments of one 16-row K block for the two column tiles of this chunk (3 planes each): requested for the first K
  // block of the slab at the top of the stage, for the second one right after the MFMAs of the first have been issued
  // (a stage whose slab holds 16 channels or fewer -- multiplicities that are not multiples of 32 -- repeats its block
  // with B = 0: branch-free, wasted work only for such shapes).  They come out of the L2 (all wavefronts of a chunk read
  // the same 12 KiB per stage) and land behind the LDS write / B split; double-buffering them (measured) buys nothing.
  nl_u32x4 Af[1][2][NPL];
  int we_blk = 0;  // F16: exponent of the K block whose fragments are in Af
  auto load_a = [&](int buf, const NodeStage& st, int k16) {
    const nl_u32x4* __restrict__ p = pa.wf + (int64_t)st.t * pa.frag_stride + pa.frag_off[st.q] + lane +
                                     ((int64_t)(k16 * nct + ct0) * NPL) * 64;
    if constexpr (F16) {
      Af[buf][0][0] = p[0]; Af[buf][0][1] = p[64];
      if (two_tiles) { Af[buf][1][0] = p[128]; Af[buf][1][1] = p[192]; }
      we_blk = pa.wexp[st.t * pa.exp_stride + a.instr[st.q].pad + k16];
    } else {
      Af[buf][0][0] = p[0]; Af[buf][0][1] = p[64]; Af[buf][0][2] = p[128];
      if (two_tiles) { Af[buf][1][0] = p[192]; Af[buf][1][1] = p[256]; Af[buf][1][2] = p[320]; }
    }
  };
  f32x16n acc0 = {0}, acc1 = {0};
  constexpr int kUnset = 1 << 20;
  int Scol = kUnset;  // F16: running exponent of this lane's column
  auto block = [&](int buf, int s, bool on) {
    // B fragment: the 8 k-values 16 s + 8 half + e of this lane's column, split in registers
    const float* __restrict__ xb = xs + zl * S + m;
    float bq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bq[e] = xb[(16 * s + 8 * half + e) * D];
    // Columns are independent in the product and a column that is never stored (lanes beyond the group's atoms) may hold
    // anything; rows beyond the slab / atoms beyond N were zero-filled when the slab was staged.  Only TYPED launches have to
    // silence columns (atoms of another type than the staged weight set's).
    if (a.n_types > 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) bq[e] = on ? bq[e] : 0.f;
    }
    if constexpr (F16) {
      const int we = we_blk;
      float mx = 0.f;  // the column's largest magnitude in this K block (its other 8 values sit in lane ^ 32)
#pragma unroll
      for (int e = 0; e < 8; ++e) mx = fmaxf(mx, fabsf(bq[e]));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      int shift = 0;
      if (mx > 0.f && mx < 3.0e38f) {
        int em;
        (void)frexpf(mx, &em);
        const int cap = 15 - em + we;
        if (cap < Scol) {
          int ns = cap - 3;
          ns = ns > we + 100 ? we + 100 : (ns < we - 100 ? we - 100 : ns);
          shift = Scol == kUnset ? 0 : Scol - ns;
          Scol = ns;
        }
      }
      if (__any(shift > 0)) {  // (rare: a block 8x above everything the column has seen)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          acc0[r] = ldexpf(acc0[r], -shift);
          acc1[r] = ldexpf(acc1[r], -shift);
        }
      }
      int q = Scol == kUnset ? 0 : Scol - we;
      q = q > 120 ? 120 : (q < -120 ? -120 : q);
      const float qs = ldexpf(1.f, q);
      nl_u32x4 Bh, Bl;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v0 = bq[2 * e] * qs, v1 = bq[2 * e + 1] * qs;
        uint32_t x, y;
        nl_split_pair_f16(v0, v1, x, y);
        Bh[e] = x; Bl[e] = y;
      }
      // three partial products per tile, small ones first; the two tiles alternate
      if (two_tiles) {
        acc0 = nl_mfma_f16(Af[buf][0][1], Bh, acc0);
        acc1 = nl_mfma_f16(Af[buf][1][1], Bh, acc1);
        acc0 = nl_mfma_f16(Af[buf][0][0], Bl, acc0);
        acc1 = nl_mfma_f16(Af[buf][1][0], Bl, acc1);
        acc0 = nl_mfma_f16(Af[buf][0][0], Bh, acc0);
        acc1 = nl_mfma_f16(Af[buf][1][0], Bh, acc1);
      } else {
        acc0 = nl_mfma_f16(Af[buf][0][1], Bh, acc0);
        acc0 = nl_mfma_f16(Af[buf][0][0], Bl, acc0);
        acc0 = nl_mfma_f16(Af[buf][0][0], Bh, acc0);
      }
      return;
    }
    nl_u32x4 Bh, Bm, Bl;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float v0 = bq[2 * e], v1 = bq[2 * e + 1];
      uint32_t x, y, zz;
      nl_split_pair(v0, v1, x, y, zz);
      Bh[e] = x; Bm[e] = y; Bl[e] = zz;
    }
    // six partial products per tile, smallest first (hi.lo, lo.hi, mid.mid, mid.hi, hi.mid, hi.hi); the two tiles
    // alternate so that consecutive MFMAs never wait for each other's accumulator
    if (two_tiles) {
      acc0 = nl_mfma_bf16(Af[buf][0][0], Bl, acc0);
      acc1 = nl_mfma_bf16(Af[buf][1][0], Bl, acc1);
      acc0 = nl_mfma_bf16(Af[buf][0][2], Bh, acc0);
      acc1 = nl_mfma_bf16(Af[buf][1][2], Bh, acc1);
      acc0 = nl_mfma_bf16(Af[buf][0][1], Bm, acc0);
      acc1 = nl_mfma_bf16(Af[buf][1][1], Bm, acc1);
      acc0 = nl_mfma_bf16(Af[buf][0][1], Bh, acc0);
      acc1 = nl_mfma_bf16(Af[buf][1][1], Bh, acc1);
      acc0 = nl_mfma_bf16(Af[buf][0][0], Bm, acc0);
      acc1 = nl_mfma_bf16(Af[buf][1][0], Bm, acc1);
      acc0 = nl_mfma_bf16(Af[buf][0][0], Bh, acc0);
      acc1 = nl_mfma_bf16(Af[buf][1][0], Bh, acc1);
    } else {
      acc0 = nl_mfma_bf16(Af[buf][0][0], Bl, acc0);
      acc0 = nl_mfma_bf16(Af[buf][0][2], Bh, acc0);
      acc0 = nl_mfma_bf16(Af[buf][0][1], Bm, acc0);
      acc0 = nl_mfma_bf16(Af[buf][0][1], Bh, acc0);
      acc0 = nl_mfma_bf16(Af[buf][0][0], Bm, acc0);
      acc0 = nl_mfma_bf16(Af[buf][0][0], Bh, acc0);
    }
  };

  int nst = 0;
  const bool tl = (a.dbg & 64) != 0;
  unsigned* tl_dst = reinterpret_cast<unsigned*>(a.out) + (int64_t)unit * 16;
  unsigned long long tl_t0 = 0;
  auto stamp = [&]() {
    if (nst < 13) {
      const unsigned long long tt = __builtin_readcyclecounter();
      if (nst == 0) tl_t0 = tt;
      if (lane == 0) tl_dst[1 + nst] = nst == 0 ? (unsigned)(tt & 0xffffffffu) : (unsigned)(tt - tl_t0);
      ++nst;
    }
  };
  if (tl) stamp();

  // one stage: slab registers -> LDS, request the next slab into the same registers, then the K blocks of the slab
  float4 xr0[XV4];
  NodeStage cur = first_stage();
  NodeStage ld = cur;  // the stage whose slab is requested next
  if constexpr (PIPE && F16) {
    nl_u32x4 Ap[2][2][2][2];  // [buffer][K block][tile][plane]
    int wep[2][2];
    auto load_a2 = [&](int b, const NodeStage& st) {
      const int k16a = st.k0 >> 4;
      const bool second = st.k0 + 16 < st.mul_in;  // (a slab of <= 16 channels repeats its block with B = 0)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const int k16 = (kb == 1 && second) ? k16a + 1 : k16a;
        const nl_u32x4* __restrict__ p = pa.wf + (int64_t)st.t * pa.frag_stride + pa.frag_off[st.q] + lane +
                                         ((int64_t)(k16 * nct + ct0) * 2) * 64;
        Ap[b][kb][0][0] = p[0]; Ap[b][kb][0][1] = p[64];
        if (two_tiles) { Ap[b][kb][1][0] = p[128]; Ap[b][kb][1][1] = p[192]; }
        // (scalar load: a vector load here would sit in the in-order vmcnt queue behind the slab prefetch)
        wep[b][kb] = pa.wexp[__builtin_amdgcn_readfirstlane(st.t * pa.exp_stride + a.instr[st.q].pad + k16)];
      }
    };
    auto blocks = [&](int b, const NodeStage& st) {
      const bool bsel = col_ok && (a.n_types == 1 || tzj == st.t);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int tl2 = 0; tl2 < 2; ++tl2) {
          Af[0][tl2][0] = Ap[b][kb][tl2][0];
          Af[0][tl2][1] = Ap[b][kb][tl2][1];
        }
        we_blk = wep[b][kb];
        block(0, kb, bsel && (st.k0 + 16 * kb < st.mul_in));
      }
    };
    if (ld.valid) { load_x(xr0, ld); load_a2(0, ld); ld = next_stage(ld); }
    // the atom type of this lane's column was requested at the top of the unit: consume the load HERE, once -- its first use
    // inside the loop would otherwise carry a vmcnt(0) into every stage
    asm volatile("" ::"v"(tzj));
    while (cur.valid) {
      store_x(xr0, cur);
      __builtin_amdgcn_sched_barrier(0);
      if (ld.valid) { load_x(xr0, ld); load_a2(1, ld); }
      __builtin_amdgcn_sched_barrier(0);
      blocks(0, cur);
      cur = ld;
      ld = next_stage(ld);
      if (!cur.valid) break;
      store_x(xr0, cur);
      __builtin_amdgcn_sched_barrier(0);
      if (ld.valid) { load_x(xr0, ld); load_a2(0, ld); }
      __builtin_amdgcn_sched_barrier(0);
      blocks(1, cur);
      cur = ld;
      ld = next_stage(ld);
    }
  } else {
  if (ld.valid) { load_x(xr0, ld); ld = next_stage(ld); }
  while (cur.valid) {
    store_x(xr0, cur);
    if (tl) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); stamp(); }
    const int k16a = cur.k0 >> 4;
    load_a(0, cur, k16a);
    __builtin_amdgcn_sched_barrier(0);
    if (ld.valid) { load_x(xr0, ld); ld = next_stage(ld); }
    __builtin_amdgcn_sched_barrier(0);
    const bool bsel = col_ok && (a.n_types == 1 || tzj == cur.t);
#pragma unroll
    for (int s16 = 0; s16 < kNLK3 / 16; ++s16) {
      const bool exists = cur.k0 + 16 * s16 < cur.mul_in;  // (wave-uniform; a missing block repeats the last one with B = 0)
      block(0, s16, bsel && exists);
      if (s16 + 1 < kNLK3 / 16) {
        __builtin_amdgcn_sched_barrier(0);
        const bool nexists = cur.k0 + 16 * (s16 + 1) < cur.mul_in;
        load_a(0, cur, nexists ? k16a + s16 + 1 : k16a);  // (same registers: the MFMAs above have read them)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    cur = next_stage(cur);
  }
  }
  if (tl) {
    stamp();
    if (lane == 0) tl_dst[0] = (unsigned)nst;
    return;
  }
  // ---- epilogue (as in node_linear_wave_unit)
  constexpr int SE = kNLW * D + P;
  constexpr bool kVecE = (SE % 4) == 0;
  constexpr int RUN4E = kNLW * D / 4;
  constexpr int XV4E = (NZT * RUN4E + 63) / 64;
  static_assert(NZT * SE <= kNLXS, "result slab too small");
  if constexpr (F16) {  // back to the true scale: 2^-S of this lane's column
    const int sb = Scol == kUnset ? 0 : Scol;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc0[r] = ldexpf(acc0[r], -sb);
      acc1[r] = ldexpf(acc1[r], -sb);
    }
  }
  if (zlr < NZT) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int wl = (r & 3) + 8 * (r >> 2) + 4 * half;
      xs[zl * SE + wl * D + m] = acc0[r];
      xs[zl * SE + (wl + 32) * D + m] = acc1[r];
    }
  }
  const int run = cw * D;
  const bool oal = ((a.dout | ch.o_off | (ch.c0 * D)) & 3) == 0;
  const bool fast = oal && cw == kNLW && zbase + NZT <= a.N && kVecE;
#pragma unroll
  for (int v = 0; v < XV4E; ++v) {
    const int idx = lane + v * 64;
    const int ez = idx / RUN4E;
    const int eo = (idx - ez * RUN4E) * 4;
    const int64_t zr = zbase + ez;  // row of the launch
    if (ez >= NZT) continue;
    const int64_t zg = atom(zr);
    const float* __restrict__ sp = xs + ez * SE + eo;
    const int64_t o = zg * a.dout + ch.o_off + (int64_t)ch.c0 * D + eo;
    if (fast) {
      float4 r = *reinterpret_cast<const float4*>(sp);
      r.x *= a.scale; r.y *= a.scale; r.z *= a.scale; r.w *= a.scale;
      if (a.addend != nullptr) {
        const float4 ad = *reinterpret_cast<const float4*>(a.addend + o);
        r.x += ad.x; r.y += ad.y; r.z += ad.z; r.w += ad.w;
      }
      *reinterpret_cast<float4*>(a.out + o) = r;
    } else if (zr < a.N && eo < run) {
      float4 r;
      if constexpr (kVecE) {
        r = *reinterpret_cast<const float4*>(sp);
      } else {
        r = make_float4(sp[0], sp[1], sp[2], sp[3]);
      }
      r.x *= a.scale; r.y *= a.scale; r.z *= a.scale; r.w *= a.scale;
      if (oal && eo + 3 < run) {
        if (a.addend != nullptr) {
          const float4 ad = *reinterpret_cast<const float4*>(a.addend + o);
          r.x += ad.x; r.y += ad.y; r.z += ad.z; r.w += ad.w;
        }
        *reinterpret_cast<float4*>(a.out + o) = r;
      } else {
        const float rv[4] = {r.x, r.y, r.z, r.w};
        for (int e = 0; e < 4; ++e)
          if (eo + e < run) a.out[o + e] = rv[e] + (a.addend != nullptr ? a.addend[o + e] : 0.f);
      }
    }
  }
}

constexpr int kNLWavesPerWG = 4;  // independent wavefronts (units) per workgroup: no barrier, only fewer dispatches

template <bool F16>
__global__ __launch_bounds__(64 * kNLWavesPerWG, 3) void node_linear_wave_bf16_kernel(const NodeLinearPackedArgs pa) {
  __shared__ __align__(16) float xs_all[kNLWavesPerWG * kNLXS];
  const NodeLinearArgs<float>& a = pa.base;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int unit = (int)blockIdx.x * kNLWavesPerWG + wv;
  if (unit >= pa.grp_begin[pa.n_groups]) return;
  float* xs = xs_all + wv * kNLXS;
  int gi = 0;
  while (gi + 1 < pa.n_groups && unit >= pa.grp_begin[gi + 1]) ++gi;
  const int local = unit - pa.grp_begin[gi];
  const int n = pa.grp_n[gi];
  const NodeChunk ch = a.chunks[pa.grp_chunk0[gi] + local % n];
  const int64_t g = (int64_t)(local / n);
  switch (ch.d) {
    case 1: node_linear_wave_bf16_unit<1, F16>(pa, ch, g, xs, unit); break;
    case 3: node_linear_wave_bf16_unit<3, F16>(pa, ch, g, xs, unit); break;
    case 5: node_linear_wave_bf16_unit<5, F16>(pa, ch, g, xs, unit); break;
    case 7: node_linear_wave_bf16_unit<7, F16>(pa, ch, g, xs, unit); break;
    case 9: node_linear_wave_bf16_unit<9, F16>(pa, ch, g, xs, unit); break;
    default: break;
  }
}

__global__ __launch_bounds__(64 * kNLWavesPerWG, 2) void node_linear_pipe_kernel(const NodeLinearPackedArgs pa) {
  __shared__ __align__(16) float xs_all[kNLWavesPerWG * kNLXS];
  const NodeLinearArgs<float>& a = pa.base;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int unit = (int)blockIdx.x * kNLWavesPerWG + wv;
  if (unit >= pa.grp_begin[pa.n_groups]) return;
  float* xs = xs_all + wv * kNLXS;
  int gi = 0;
  while (gi + 1 < pa.n_groups && unit >= pa.grp_begin[gi + 1]) ++gi;
  const int local = unit - pa.grp_begin[gi];
  const int n = pa.grp_n[gi];
  const NodeChunk ch = a.chunks[pa.grp_chunk0[gi] + local % n];
  const int64_t g = (int64_t)(local / n);
  switch (ch.d) {
    case 1: node_linear_wave_bf16_unit<1, true, true>(pa, ch, g, xs, unit); break;
    case 3: node_linear_wave_bf16_unit<3, true, true>(pa, ch, g, xs, unit); break;
    case 5: node_linear_wave_bf16_unit<5, true, true>(pa, ch, g, xs, unit); break;
    case 7: node_linear_wave_bf16_unit<7, true, true>(pa, ch, g, xs, unit); break;
    case 9: node_linear_wave_bf16_unit<9, true, true>(pa, ch, g, xs, unit); break;
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

template <typename T>
__device__ __forceinline__ T act_grad2(int act, T x, T cst) {
  if (act == 1) {  // d2/dx2 of x sigma(x)
    const T s = T(1) / (T(1) + exp(-x));
    return cst * s * (T(1) - s) * (T(2) + x * (T(1) - T(2) * s));
  }
  if (act == 2) {
    const T t = tanh(x);
    return -T(2) * cst * t * (T(1) - t * t);
  }
  return T(0);
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
  const T* __restrict__ gout;  // backward (and second order) only
  const T* __restrict__ cot;   // second order only: cotangent of grad_in [N, din]
  T* __restrict__ out;         // fwd: out [N, dout]; bwd: gin [N, din]; 2: [N, dout]; 3: [N, din]
  const GateCol* __restrict__ cols;
  int32_t din, dout;
  int64_t N;
};

constexpr int kGateMaxD = 9;   // 2 l + 1 for l <= 4
constexpr int kGateAtoms = 2;  // atoms per thread (unrolled: all their loads in flight at once).  16 serial atoms per
// thread left the launch latency-bound (12 / 32 us for the 10 125-atom cfg-3 rows, 51 MB of traffic)

template <typename T>
__global__ __launch_bounds__(256) void gate_fwd_kernel(const GateArgs<T> a) {
  const int c = blockIdx.y * blockDim.x + threadIdx.x;  // output column
  if (c >= a.dout) return;
  const GateCol t = a.cols[c];
  const T cst = (T)t.cst;
  const int64_t z0 = (int64_t)blockIdx.x * kGateAtoms;
  const int64_t z1 = min(z0 + kGateAtoms, a.N);
#pragma unroll
  for (int64_t z = z0; z < z1; ++z) {
    const T* __restrict__ row = a.in + z * a.din;
    const T xs = row[t.a];
    a.out[z * a.dout + c] = t.b < 0 ? act_eval(t.c, xs, cst) : act_eval(t.c, row[t.b], cst) * xs;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gate_bwd_kernel(const GateArgs<T> a) {
  const int c = blockIdx.y * blockDim.x + threadIdx.x;  // input column
  if (c >= a.din) return;
  const GateCol t = a.cols[c];
  const T cst = (T)t.cst;
  const int64_t z0 = (int64_t)blockIdx.x * kGateAtoms;
  const int64_t z1 = min(z0 + kGateAtoms, a.N);
#pragma unroll
  for (int64_t z = z0; z < z1; ++z) {
    const T* __restrict__ row = a.in + z * a.din;
    const T* __restrict__ g = a.gout + z * a.dout;
    T r;
    if (t.a == 0) {
      r = g[t.c] * act_grad(t.b, row[c], cst);
    } else if (t.a == 1) {
      // all 2 * len loads requested at once (a runtime-bounded loop issues load, wait, fma per component: the gate
      // columns then take len serial memory round trips and the whole launch waits for them)
      T s = T(0);
#pragma unroll
      for (int m = 0; m < kGateMaxD; ++m) {
        const int mm = m < t.e ? m : 0;
        const T v = g[t.c + mm] * row[t.d + mm];
        s += m < t.e ? v : T(0);
      }
      r = s * act_grad(t.b, row[c], cst);
    } else if (t.a == 2) {
      r = act_eval(t.b, row[t.f], cst) * g[t.c];
    } else {
      r = T(0);
    }
    a.out[z * a.din + c] = r;
  }
}

// Second order (force-matching training differentiates the backward pass): with gin = gate_bwd(x, g) and a cotangent
// c [N, din] of gin,
//   mode 2 (per OUTPUT column, forward table): d<c, gin>/dg
//       scalar o <- column s:  c_s a'(x_s);      gated o <- (v column i, gate column q):  c_q a'(x_q) v_i + c_i a(x_q)
//   mode 3 (per INPUT column, backward table): d<c, gin>/dx
//       scalar s:  c_s g_o a''(x_s);   gate q:  c_q a''(x_q) sum_m g_{o+m} v_{i+m} + a'(x_q) sum_m c_{i+m} g_{o+m};
//       gated v_i:  c_q a'(x_q) g_o
template <typename T>
__global__ __launch_bounds__(256) void gate_bwd_bwd_g_kernel(const GateArgs<T> a) {
  const int c = blockIdx.y * blockDim.x + threadIdx.x;  // output column
  if (c >= a.dout) return;
  const GateCol t = a.cols[c];
  const T cst = (T)t.cst;
  const int64_t z0 = (int64_t)blockIdx.x * kGateAtoms;
  const int64_t z1 = min(z0 + kGateAtoms, a.N);
#pragma unroll
  for (int64_t z = z0; z < z1; ++z) {
    const T* __restrict__ row = a.in + z * a.din;
    const T* __restrict__ ct = a.cot + z * a.din;
    T r;
    if (t.b < 0) {
      r = ct[t.a] * act_grad(t.c, row[t.a], cst);
    } else {
      const T xq = row[t.b];
      r = ct[t.b] * act_grad(t.c, xq, cst) * row[t.a] + ct[t.a] * act_eval(t.c, xq, cst);
    }
    a.out[z * a.dout + c] = r;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gate_bwd_bwd_x_kernel(const GateArgs<T> a) {
  const int c = blockIdx.y * blockDim.x + threadIdx.x;  // input column
  if (c >= a.din) return;
  const GateCol t = a.cols[c];
  const T cst = (T)t.cst;
  const int64_t z0 = (int64_t)blockIdx.x * kGateAtoms;
  const int64_t z1 = min(z0 + kGateAtoms, a.N);
#pragma unroll
  for (int64_t z = z0; z < z1; ++z) {
    const T* __restrict__ row = a.in + z * a.din;
    const T* __restrict__ g = a.gout + z * a.dout;
    const T* __restrict__ ct = a.cot + z * a.din;
    T r;
    if (t.a == 0) {
      r = ct[c] * g[t.c] * act_grad2(t.b, row[c], cst);
    } else if (t.a == 1) {
      T sgv = T(0), scg = T(0);
#pragma unroll
      for (int m = 0; m < kGateMaxD; ++m) {
        const int mm = m < t.e ? m : 0;
        const T gm = g[t.c + mm];
        sgv += m < t.e ? gm * row[t.d + mm] : T(0);
        scg += m < t.e ? ct[t.d + mm] * gm : T(0);
      }
      r = ct[c] * act_grad2(t.b, row[c], cst) * sgv + act_grad(t.b, row[c], cst) * scg;
    } else if (t.a == 2) {
      r = ct[t.f] * act_grad(t.b, row[t.f], cst) * g[t.c];
    } else {
      r = T(0);
    }
    a.out[z * a.din + c] = r;
  }
}

}  // namespace nqa

using namespace nqa;

extern "C" {

int nqa_node_linear(int32_t dtype, const void* x, const void* weights, const void* addend, void* out,
                    const int64_t* atom_types, const void* chunk_table, int32_t n_chunks, const void* instr_table,
                    int32_t n_instr, int32_t n_types, int64_t weight_stride, int32_t dim_in, int32_t dim_out, int64_t num_nodes,
                    double scale, int32_t chunk_width, nqa_stream stream) {
  return nqa_node_linear_ordered(dtype, x, weights, addend, out, atom_types, nullptr, chunk_table, n_chunks, instr_table,
                                 n_instr, n_types, weight_stride, dim_in, dim_out, num_nodes, scale, chunk_width, stream);
}

int nqa_node_linear_ordered(int32_t dtype, const void* x, const void* weights, const void* addend, void* out,
                            const int64_t* atom_types, const int32_t* atom_order, const void* chunk_table, int32_t n_chunks,
                            const void* instr_table, int32_t n_instr, int32_t n_types, int64_t weight_stride,
                            int32_t dim_in, int32_t dim_out, int64_t num_nodes, double scale, int32_t chunk_width,
                            nqa_stream stream) {
  // 64-channel chunk tables serve both kernels: float32 runs on fp32 MFMA (chunk_width 64) or, on request
  // (chunk_width -64), on the VALU kernel that also serves float64
  const bool use_mfma = dtype == NQA_F32 && chunk_width == 64;
  if (chunk_width != 64 && chunk_width != -64) {
    set_error("nqa_node_linear: chunk_width must be 64 (or -64: VALU kernel)");
    return NQA_ERR_INVALID;
  }
  if (dtype != NQA_F32 && dtype != NQA_F64) {
    set_error("nqa_node_linear: unsupported dtype");
    return NQA_ERR_UNSUPPORTED;
  }
  if (n_instr > kMaxNodeInstr) {
    set_error("nqa_node_linear: more than 64 instructions in one call");
    return NQA_ERR_UNSUPPORTED;
  }
  if (n_chunks > kMaxNodeChunks && chunk_table != nullptr) {
    // wide layers (e.g. l_max = 3 with 128 features: > 40 output chunks): one launch per group of chunks -- chunk
    // records are self-contained (absolute offsets / instruction ranges)
    for (int32_t c0 = 0; c0 < n_chunks; c0 += kMaxNodeChunks) {
      const int32_t nc = n_chunks - c0 < kMaxNodeChunks ? n_chunks - c0 : kMaxNodeChunks;
      const int rc = nqa_node_linear_ordered(dtype, x, weights, addend, out, atom_types, atom_order,
                                             static_cast<const NodeChunk*>(chunk_table) + c0, nc, instr_table, n_instr,
                                             n_types, weight_stride, dim_in, dim_out, num_nodes, scale, chunk_width, stream);
      if (rc != NQA_OK) return rc;
    }
    return NQA_OK;
  }
  if (num_nodes < 0 || n_chunks < 0 || n_instr < 0 || n_types < 1 || dim_in <= 0 || dim_out <= 0 ||
      (num_nodes > 0 && (!x || !weights || !out || !chunk_table || (n_instr > 0 && !instr_table))) ||
      (n_types > 1 && atom_types == nullptr)) {
    set_error("nqa_node_linear: invalid argument");
    return NQA_ERR_INVALID;
  }
  if (num_nodes == 0) return NQA_OK;
  const size_t es = dtype == NQA_F32 ? 4 : 8;
  const size_t smem = (size_t)kNZ * dim_in * es;
  if (!use_mfma && smem > 160 * 1024 - 1024) {  // (the MFMA kernel stages 64-channel slabs, not whole rows)
    set_error("nqa_node_linear: feature rows too wide for the LDS tile of the VALU kernel");
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
    a.perm = use_mfma ? atom_order : nullptr;  // (the VALU kernel keeps the natural order)
    std::memcpy(a.chunks, chunk_table, sizeof(NodeChunk) * (size_t)n_chunks);
    if (n_instr > 0) std::memcpy(a.instr, instr_table, sizeof(NodeInstr) * (size_t)n_instr);
    a.n_chunks = n_chunks;
    a.n_types = n_types;
    a.din = dim_in;
    a.dout = dim_out;
    a.wstride = weight_stride;
    a.N = num_nodes;
    a.scale = (float)scale;
    {
      static const int dbg = [] {
        const char* e = std::getenv("NQA_NODE_DBG");
        return e ? std::atoi(e) : 0;
      }();
      a.dbg = dbg;
    }
    if (use_mfma) {
      // per-wavefront kernel: one 64-thread workgroup per (chunk, floor(32/d) atoms) unit; chunks with the most stages
      // (instruction x type x 32-channel K slab) first, so that the long units are dispatched first
      int order[kMaxNodeChunks];
      int stages[kMaxNodeChunks];
      for (int c = 0; c < n_chunks; ++c) {
        const int d = a.chunks[c].d;
        if (d != 1 && d != 3 && d != 5 && d != 7 && d != 9) {
          set_error("nqa_node_linear: irrep dimension above 9 (l > 4)");
          return NQA_ERR_UNSUPPORTED;
        }
        int st = 0;
        for (int q = a.chunks[c].instr_begin; q < a.chunks[c].instr_end; ++q)
          st += n_types * ((a.instr[q].mul_in + kNLK2 - 1) / kNLK2);
        stages[c] = st;
        order[c] = c;
      }
      std::stable_sort(order, order + n_chunks, [&](int l, int r) { return stages[l] > stages[r]; });
      NodeChunk sorted[kMaxNodeChunks];
      for (int c = 0; c < n_chunks; ++c) sorted[c] = a.chunks[order[c]];
      int64_t nblk = 0;
      for (int c = 0; c < n_chunks; ++c) {
        a.chunks[c] = sorted[c];
        a.blk_begin[c] = (int32_t)nblk;
        const int per_unit = 32 / a.chunks[c].d;
        nblk += (num_nodes + per_unit - 1) / per_unit;
      }
      a.blk_begin[n_chunks] = (int32_t)nblk;
      if (nblk > 2147483647LL) {
        set_error("nqa_node_linear: too many work units for one launch");
        return NQA_ERR_UNSUPPORTED;
      }
      if (nblk == 0) return NQA_OK;
      hipLaunchKernelGGL(node_linear_wave_kernel, dim3((unsigned)nblk), dim3(64), 0, s, a);
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
    std::memcpy(a.chunks, chunk_table, sizeof(NodeChunk) * (size_t)n_chunks);
    if (n_instr > 0) std::memcpy(a.instr, instr_table, sizeof(NodeInstr) * (size_t)n_instr);
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

// ---- packed (split-bf16) weights ------------------------------------------------------------------------------------
namespace {
// fragment layout of one atom type: per instruction ceil(mul_in / 16) x ceil(mul_out / 32) fragments of 3 planes x 64 lanes
// (uint4 each); mul_out is that of the output block the instruction feeds.  Returns the uint4 count per type, -1 on error.
// two-plane fp16 split with running column scales (default) or the three-plane bf16 split (NQA_NODE_F16=0).  Read at every
// call: a packed buffer must be used in the mode it was packed in (the Python host keys its cache on the variable).
bool node_f16() {
  const char* e = std::getenv("NQA_NODE_F16");
  return e == nullptr || e[0] != '0';
}

// exponent table of the F16 layout: one int per (instruction, 16-row K block); returns the count per type
int32_t node_exp_layout(const nqa::NodeInstr* instr, int32_t n_instr, int32_t* exp_off) {
  int32_t off = 0;
  for (int q = 0; q < n_instr; ++q) {
    exp_off[q] = off;
    off += (instr[q].mul_in + 15) / 16;
  }
  exp_off[n_instr] = off;
  return off;
}

int64_t node_frag_layout(const nqa::NodeChunk* chunks, int32_t n_chunks, const nqa::NodeInstr* instr, int32_t n_instr,
                         int32_t* frag_off, int32_t* mul_out, int planes = node_f16() ? 2 : 3) {
  for (int q = 0; q < n_instr; ++q) mul_out[q] = 0;
  for (int c = 0; c < n_chunks; ++c)
    for (int q = chunks[c].instr_begin; q < chunks[c].instr_end; ++q) {
      if (q < 0 || q >= n_instr) return -1;
      mul_out[q] = chunks[c].mul_out;
    }
  int64_t off = 0;
  for (int q = 0; q < n_instr; ++q) {
    frag_off[q] = (int32_t)off;
    off += (int64_t)((instr[q].mul_in + 15) / 16) * ((mul_out[q] + 31) / 32) * planes * 64;
    if (off > 2147483647LL) return -1;
  }
  frag_off[n_instr] = (int32_t)off;
  return off;
}
}  // namespace

int64_t nqa_node_weights_pack_bytes(const void* chunk_table, int32_t n_chunks, const void* instr_table, int32_t n_instr,
                                    int32_t n_types) {
  if (n_instr < 0 || n_instr > nqa::kMaxNodeInstr || n_chunks < 0 || n_types < 1 || (n_chunks > 0 && !chunk_table) ||
      (n_instr > 0 && !instr_table))
    return -1;
  int32_t frag_off[nqa::kMaxNodeInstr + 1], mul_out[nqa::kMaxNodeInstr];
  const int64_t per_type = node_frag_layout(static_cast<const nqa::NodeChunk*>(chunk_table), n_chunks,
                                            static_cast<const nqa::NodeInstr*>(instr_table), n_instr, frag_off, mul_out);
  if (per_type < 0) return -1;
  int32_t exp_off[nqa::kMaxNodeInstr + 1];
  const int64_t nexp = node_f16() ? node_exp_layout(static_cast<const nqa::NodeInstr*>(instr_table), n_instr, exp_off) : 0;
  return per_type * 16 * n_types + ((nexp * n_types * 4 + 255) & ~(int64_t)255);
}

int nqa_node_weights_pack(const void* weights, const void* chunk_table, int32_t n_chunks, const void* instr_table,
                          int32_t n_instr, int32_t n_types, int64_t weight_stride, void* packed, nqa_stream stream) {
  using namespace nqa;
  if (n_instr < 0 || n_instr > kMaxNodeInstr || n_chunks < 0 || n_types < 1 || !weights || !packed ||
      (n_chunks > 0 && !chunk_table) || (n_instr > 0 && !instr_table)) {
    set_error("nqa_node_weights_pack: invalid argument");
    return NQA_ERR_INVALID;
  }
  NodePackArgs a{};
  const NodeInstr* instr = static_cast<const NodeInstr*>(instr_table);
  const int64_t per_type = node_frag_layout(static_cast<const NodeChunk*>(chunk_table), n_chunks, instr, n_instr,
                                            a.frag_off, a.mul_out);
  if (per_type < 0) {
    set_error("nqa_node_weights_pack: inconsistent tables");
    return NQA_ERR_INVALID;
  }
  if (per_type == 0) return NQA_OK;
  for (int q = 0; q < n_instr; ++q) {
    a.mul_in[q] = instr[q].mul_in;
    a.w_off[q] = instr[q].w_off;
  }
  a.w = static_cast<const float*>(weights);
  a.out = static_cast<nl_u32x4*>(packed);
  a.wstride = weight_stride;
  a.frag_stride = per_type;
  a.n_instr = n_instr;
  a.n_types = n_types;
  if (node_f16()) {
    a.exp_stride = node_exp_layout(instr, n_instr, a.exp_off);
    a.wexp = reinterpret_cast<int32_t*>(static_cast<char*>(packed) + per_type * 16 * n_types);
    const int64_t nexp = (int64_t)a.exp_stride * n_types;
    hipLaunchKernelGGL(node_weights_exp_kernel, dim3((unsigned)nexp), dim3(64), 0, static_cast<hipStream_t>(stream), a);
    const int64_t threads = per_type / 2 * n_types;
    hipLaunchKernelGGL(node_weights_pack_f16_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a);
  } else {
    const int64_t threads = per_type / 3 * n_types;
    hipLaunchKernelGGL(node_weights_pack_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a);
  }
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_error(std::string("nqa_node_weights_pack: ") + hipGetErrorString(err));
    return NQA_ERR_LAUNCH;
  }
  return NQA_OK;
}

int nqa_node_linear_packed(const void* x, const void* packed, const void* addend, void* out, const int64_t* atom_types,
                           const void* chunk_table, int32_t n_chunks, const void* instr_table, int32_t n_instr,
                           int32_t n_types, int32_t dim_in, int32_t dim_out, int64_t num_nodes, double scale,
                           nqa_stream stream) {
  return nqa_node_linear_packed_ordered(x, packed, addend, out, atom_types, nullptr, chunk_table, n_chunks, instr_table,
                                        n_instr, n_types, dim_in, dim_out, num_nodes, scale, stream);
}

int nqa_node_linear_packed_ordered(const void* x, const void* packed, const void* addend, void* out,
                                   const int64_t* atom_types, const int32_t* atom_order, const void* chunk_table,
                                   int32_t n_chunks, const void* instr_table, int32_t n_instr, int32_t n_types,
                                   int32_t dim_in, int32_t dim_out, int64_t num_nodes, double scale, nqa_stream stream) {
  using namespace nqa;
  if (n_instr < 0 || n_instr > kMaxNodeInstr || num_nodes < 0 || n_chunks < 0 || n_types < 1 || dim_in <= 0 ||
      dim_out <= 0 || (num_nodes > 0 && (!x || !packed || !out || !chunk_table || (n_instr > 0 && !instr_table))) ||
      (n_types > 1 && atom_types == nullptr)) {
    set_error("nqa_node_linear_packed: invalid argument");
    return NQA_ERR_INVALID;
  }
  if (num_nodes == 0 || n_chunks == 0) return NQA_OK;
  const NodeChunk* chunks = static_cast<const NodeChunk*>(chunk_table);
  const NodeInstr* instr = static_cast<const NodeInstr*>(instr_table);
  NodeLinearPackedArgs pa{};
  int32_t frag_off[kMaxNodeInstr + 1], mul_out[kMaxNodeInstr];
  const int64_t per_type = node_frag_layout(chunks, n_chunks, instr, n_instr, frag_off, mul_out);
  if (per_type < 0) {
    set_error("nqa_node_linear_packed: inconsistent tables");
    return NQA_ERR_INVALID;
  }
  for (int q = 0; q < n_instr; ++q) pa.frag_off[q] = frag_off[q];
  pa.frag_stride = per_type;
  pa.wf = static_cast<const nl_u32x4*>(packed);
  int32_t exp_off[kMaxNodeInstr + 1];
  pa.exp_stride = node_exp_layout(instr, n_instr, exp_off);
  pa.wexp = reinterpret_cast<const int32_t*>(static_cast<const char*>(packed) + per_type * 16 * n_types);
  NodeLinearArgs<float>& a = pa.base;
  a.x = static_cast<const float*>(x);
  a.addend = static_cast<const float*>(addend);
  a.out = static_cast<float*>(out);
  a.types = n_types > 1 ? atom_types : nullptr;
  a.perm = atom_order;
  if (n_instr > 0) std::memcpy(a.instr, instr, sizeof(NodeInstr) * (size_t)n_instr);
  for (int q = 0; q < n_instr; ++q) a.instr[q].pad = exp_off[q];  // (F16: first exponent of the instruction)
  a.n_types = n_types;
  a.din = dim_in;
  a.dout = dim_out;
  a.N = num_nodes;
  a.scale = (float)scale;
  {
    static const int dbg = [] {
      const char* e = std::getenv("NQA_NODE_DBG");
      return e ? std::atoi(e) : 0;
    }();
    a.dbg = dbg;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  // chunks with the most stages first; more than kMaxNodeChunks chunks (wide l_max = 3 layers) in several launches
  std::vector<int> order(n_chunks), stages(n_chunks);
  for (int c = 0; c < n_chunks; ++c) {
    const int d = chunks[c].d;
    if (d != 1 && d != 3 && d != 5 && d != 7 && d != 9) {
      set_error("nqa_node_linear_packed: irrep dimension above 9 (l > 4)");
      return NQA_ERR_UNSUPPORTED;
    }
    if (chunks[c].c0 % 64 != 0) {
      set_error("nqa_node_linear_packed: chunks must start at multiples of 64 channels");
      return NQA_ERR_INVALID;
    }
    int st = 0;
    for (int q = chunks[c].instr_begin; q < chunks[c].instr_end; ++q) st += n_types * ((instr[q].mul_in + kNLK3 - 1) / kNLK3);
    stages[c] = st;
    order[c] = c;
  }
  std::stable_sort(order.begin(), order.end(), [&](int l, int r) { return stages[l] > stages[r]; });
  for (int c0 = 0; c0 < n_chunks; c0 += kMaxNodeChunks) {
    const int nc = std::min(n_chunks - c0, kMaxNodeChunks);
    int64_t nblk = 0;
    int ng = 0;
    for (int c = 0; c < nc; ++c) {
      a.chunks[c] = chunks[order[c0 + c]];
      const NodeChunk& cc = a.chunks[c];
      const bool same = c > 0 && cc.instr_begin == a.chunks[c - 1].instr_begin &&
                        cc.instr_end == a.chunks[c - 1].instr_end && cc.d == a.chunks[c - 1].d &&
                        cc.o_off == a.chunks[c - 1].o_off;
      const int per_unit = 32 / cc.d;
      if (!same) {
        pa.grp_begin[ng] = (int32_t)nblk;
        pa.grp_chunk0[ng] = c;
        pa.grp_n[ng] = 0;
        ++ng;
      }
      ++pa.grp_n[ng - 1];
      nblk += (num_nodes + per_unit - 1) / per_unit;
    }
    pa.grp_begin[ng] = (int32_t)nblk;
    pa.n_groups = ng;
    a.n_chunks = nc;
    if (nblk > 2147483647LL) {
      set_error("nqa_node_linear_packed: too many work units for one launch");
      return NQA_ERR_UNSUPPORTED;
    }
    // NQA_NODE_PIPE=1: the one-wait-per-stage loop (two wavefronts per SIMD); measured no faster than the default, see
    // node_fused.h
    const char* pe = std::getenv("NQA_NODE_PIPE");
    const bool pipe = pe != nullptr && pe[0] == '1';
    if (node_f16() && pipe && (a.dbg & 64) == 0)
      hipLaunchKernelGGL(node_linear_pipe_kernel, dim3((unsigned)((nblk + kNLWavesPerWG - 1) / kNLWavesPerWG)),
                         dim3(64 * kNLWavesPerWG), 0, s, pa);
    else if (node_f16())
      hipLaunchKernelGGL(node_linear_wave_bf16_kernel<true>, dim3((unsigned)((nblk + kNLWavesPerWG - 1) / kNLWavesPerWG)),
                         dim3(64 * kNLWavesPerWG), 0, s, pa);
    else
      hipLaunchKernelGGL(node_linear_wave_bf16_kernel<false>, dim3((unsigned)((nblk + kNLWavesPerWG - 1) / kNLWavesPerWG)),
                         dim3(64 * kNLWavesPerWG), 0, s, pa);
  }
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_error(std::string("nqa_node_linear_packed: ") + hipGetErrorString(err));
    return NQA_ERR_LAUNCH;
  }
  return NQA_OK;
}

int nqa_gate(int32_t dtype, int32_t backward, const void* input, const void* grad_out, const void* cotangent,
             void* out, const void* col_table, int32_t dim_in, int32_t dim_out, int64_t num_nodes, nqa_stream stream) {
  if (dtype != NQA_F32 && dtype != NQA_F64) {
    set_error("nqa_gate: unsupported dtype");
    return NQA_ERR_UNSUPPORTED;
  }
  if (backward < 0 || backward > 3 || num_nodes < 0 ||
      (num_nodes > 0 && (!input || !out || !col_table || ((backward == 1 || backward == 3) && !grad_out) ||
                         (backward >= 2 && !cotangent))) ||
      dim_in <= 0 || dim_out <= 0) {
    set_error("nqa_gate: invalid argument");
    return NQA_ERR_INVALID;
  }
  if (num_nodes == 0) return NQA_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int cols = (backward == 1 || backward == 3) ? dim_in : dim_out;
  const int64_t ny = (num_nodes + kGateAtoms - 1) / kGateAtoms;
  if (ny > 2147483647LL) {
    set_error("nqa_gate: too many atoms for one launch");
    return NQA_ERR_UNSUPPORTED;
  }
  const unsigned bdim = cols >= 256 ? 256u : (unsigned)(((cols + 63) / 64) * 64);
  const dim3 grid((unsigned)ny, (unsigned)((cols + bdim - 1) / bdim));
#define NQA_GATE_LAUNCH(T)                                                                        \
  {                                                                                               \
    GateArgs<T> a{};                                                                              \
    a.in = static_cast<const T*>(input);                                                          \
    a.gout = static_cast<const T*>(grad_out);                                                     \
    a.cot = static_cast<const T*>(cotangent);                                                     \
    a.out = static_cast<T*>(out);                                                                 \
    a.cols = static_cast<const GateCol*>(col_table);                                              \
    a.din = dim_in;                                                                               \
    a.dout = dim_out;                                                                             \
    a.N = num_nodes;                                                                              \
    if (backward == 1)                                                                            \
      hipLaunchKernelGGL(gate_bwd_kernel<T>, grid, dim3(bdim), 0, s, a);                          \
    else if (backward == 2)                                                                       \
      hipLaunchKernelGGL(gate_bwd_bwd_g_kernel<T>, grid, dim3(bdim), 0, s, a);                    \
    else if (backward == 3)                                                                       \
      hipLaunchKernelGGL(gate_bwd_bwd_x_kernel<T>, grid, dim3(bdim), 0, s, a);                    \
    else                                                                                          \
      hipLaunchKernelGGL(gate_fwd_kernel<T>, grid, dim3(bdim), 0, s, a);                          \
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

#include "node_fused.h"
