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

#include <cstdint>
#include <cstring>
#include <cstdlib>
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

constexpr int kMaxNodeChunks = 40;  // 64-channel chunks of the output irreps per launch
constexpr int kMaxNodeInstr = 64;   // (input block -> output block) matrices per launch

template <typename T>
struct NodeLinearArgs {
  const T* __restrict__ x;
  const T* __restrict__ w;       // [n_types][wstride]
  const T* __restrict__ addend;  // optional [N, dout]
  T* __restrict__ out;
  const int64_t* __restrict__ types;  // optional [N]
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
// Transposed product  D[i = w][j = (z, m)] = sum_u A[i][u] B[u][j]  on v_mfma_f32_32x32x2_f32 (exact fp32):
//   A[i = w][k = u]   = W_t[u][c0 + w]          weight slab, staged once per workgroup in LDS
//   B[k = u][j = z,m] = x[z, x_off + u*d + m]    node-row slab, staged per wavefront in LDS
// A workgroup owns one 64-channel chunk of one output irrep block; each of its 4 wavefronts owns floor(32/d) atoms
// (their d components fill the 32 MFMA columns).  The loop runs over "stages" = (instruction, atom type, 64-channel
// K slab): a stage's weight slab [64][64] is fetched coalesced by the whole workgroup (double-buffered in LDS,
// requested one stage ahead into registers), its x slab -- for every atom one *contiguous* run of 64*d floats --
// coalesced by the owning wavefront.  All MFMA operands then come from LDS with conflict-free strides, so HBM/L2
// only ever sees full-line requests (the first version read x and wrote out as scattered dwords: 3-4x off the
// roofline).  The result tile goes back through the wavefront's LDS slab and leaves as contiguous 64*d-float runs
// per atom, fused with the scale and the optional addend (self-connection / residual).  Ragged edges (odd mul,
// partial atom groups) are handled by zero-filled slabs and masked stores; for the per-type self-connection the
// columns of atoms whose type differs from the staged weight set are zeroed in the B operand.
using f32x16n = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ void nl_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int kNLW = 64;                       // output channels per chunk (two 32-row MFMA tiles)
constexpr int kNLK = 64;                       // input channels per stage
constexpr int kNLXS = 32 * (kNLK + 1);         // floats per wavefront slab: NZT atoms x (64+1)*d, NZT*d <= 32

template <int D>
__device__ __forceinline__ void node_linear_mfma_block(const NodeLinearArgs<float>& a, const NodeChunk& ch, int bx,
                                                       int nblk, float* __restrict__ ws, float* __restrict__ xs_all) {
  constexpr int NZT = 32 / D;                               // atoms per wavefront
  // padded slab stride per atom.  Columns (zl, m) of a k-step read xs[zl*S + u*d + m]: conflict-free iff zl*S + m
  // are distinct mod 32.  d > 1: S = 64*d + P with P the multiple of 4 >= d (also keeps rows 16-byte aligned for
  // 128-bit LDS access); d = 1 (32 atoms): S = 65.
  constexpr int P = D == 1 ? 1 : ((D + 3) / 4) * 4;
  constexpr int S = kNLK * D + P;
  constexpr bool kVecLds = (S % 4) == 0;
  constexpr int RUN4 = kNLK * D / 4;                        // float4 per atom run (64*d floats)
  constexpr int XV4 = (NZT * RUN4 + 63) / 64;               // float4 per lane per slab
  static_assert(NZT * S <= kNLXS, "slab too small");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, j = lane & 31;
  const int zlr = j / D, m = j - zlr * D;
  const int zl = min(zlr, NZT - 1);                         // clamped for addressing
  const int cw = min(kNLW, ch.mul_out - ch.c0);
  // atom groups (4 wavefronts x NZT atoms) of this chunk are dealt round-robin to its nblk workgroups: a workgroup
  // that owns several groups requests the first slab of the next group before the last MFMAs of the current one, so
  // only its very first operand fetch is exposed
  const int64_t ngroups = (a.N + 4 * NZT - 1) / (4 * NZT);
  float* __restrict__ xs = xs_all + wv * kNLXS;

  // per-lane slab coordinates (loop invariant): float4 v of this lane belongs to atom xz[v], offset 4*xo4[v]
  int xz[XV4], xo[XV4];
#pragma unroll
  for (int v = 0; v < XV4; ++v) {
    const int idx = lane + v * 64;
    xz[v] = idx / RUN4;
    xo[v] = (idx - xz[v] * RUN4) * 4;
  }

  // stage enumeration: (instruction q, type t, K slab k0)
  int q = ch.instr_begin, t = 0, k0 = 0;
  NodeInstr ins = q < ch.instr_end ? a.instr[q] : NodeInstr{0, 0, 0, 0};
  auto advance = [&]() {  // -> false when the stages of one atom group are exhausted
    k0 += kNLK;
    if (k0 >= ins.mul_in) {
      k0 = 0;
      if (++t >= a.n_types) {
        t = 0;
        if (++q < ch.instr_end) ins = a.instr[q];
      }
    }
    return q < ch.instr_end;
  };

  float4 wreg[4];
  float4 xreg[XV4];
  // lanes of float4 slot v that map to a real atom run: all of them except possibly in the last slot
  auto slot_ok = [&](int v) { return (v + 1) * 64 <= NZT * RUN4 || xz[v] < NZT; };
  auto load_stage = [&](const NodeInstr& si, int st, int sk0, int64_t zbase, bool with_w) {
    // weight slab rows u = sk0 .. sk0+63, columns c0 .. c0+63 of W_t [mul_in][mul_out]
    const float* __restrict__ wb = a.w + (int64_t)st * a.wstride + si.w_off + ch.c0;
    const bool wal = ((ch.mul_out | ch.c0 | si.w_off) & 3) == 0 && (a.wstride & 3) == 0;
    if (!with_w) {
      // single-stage chunk: the weight slab staged for the first atom group serves all of them
    } else if (wal && cw == kNLW && sk0 + kNLK <= si.mul_in) {
      // common case (wave-uniform): full aligned slab, no per-element predication
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int idx = tid + v * 256;  // float4 index in the [64][16] slab
        wreg[v] = *reinterpret_cast<const float4*>(wb + (int64_t)(sk0 + (idx >> 4)) * ch.mul_out + (idx & 15) * 4);
      }
    } else {
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int idx = tid + v * 256;
        const int u = idx >> 4, c4 = (idx & 15) * 4;
        const int ug = sk0 + u;
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ug < si.mul_in) {
          const float* __restrict__ p = wb + (int64_t)ug * ch.mul_out + c4;
          if (wal && c4 + 3 < cw) {
            r = *reinterpret_cast<const float4*>(p);
          } else {
            if (c4 + 0 < cw) r.x = p[0];
            if (c4 + 1 < cw) r.y = p[1];
            if (c4 + 2 < cw) r.z = p[2];
            if (c4 + 3 < cw) r.w = p[3];
          }
        }
        wreg[v] = r;
      }
    }
    // x slab: atom zz contributes the contiguous run x[zz, x_off + sk0*d .. + 64*d) (zero beyond mul_in)
    const int kk = min(kNLK, si.mul_in - sk0) * D;  // valid floats per atom
    const bool xal = ((a.din | si.x_off) & 3) == 0;  // (sk0*d is a multiple of 64)
    if (xal && kk == kNLK * D && zbase + NZT <= a.N) {
      // common case (wave-uniform): aligned full runs of a complete atom group
      const float* __restrict__ xb0 = a.x + zbase * a.din + si.x_off + sk0 * D;
#pragma unroll
      for (int v = 0; v < XV4; ++v) {
        if (slot_ok(v)) xreg[v] = *reinterpret_cast<const float4*>(xb0 + (int64_t)xz[v] * a.din + xo[v]);
      }
    } else {
#pragma unroll
      for (int v = 0; v < XV4; ++v) {
        const int64_t zg = zbase + xz[v];
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (xz[v] < NZT && zg < a.N && xo[v] < kk) {
          const float* __restrict__ p = a.x + zg * a.din + si.x_off + sk0 * D + xo[v];
          if (xal && xo[v] + 3 < kk) {
            r = *reinterpret_cast<const float4*>(p);
          } else {
            r.x = p[0];
            if (xo[v] + 1 < kk) r.y = p[1];
            if (xo[v] + 2 < kk) r.z = p[2];
            if (xo[v] + 3 < kk) r.w = p[3];
          }
        }
        xreg[v] = r;
      }
    }
  };
  auto store_stage = [&](int buf, bool with_w) {
    if (with_w) {
#pragma unroll
      for (int v = 0; v < 4; ++v) *reinterpret_cast<float4*>(ws + buf * (kNLK * kNLW) + (tid + v * 256) * 4) = wreg[v];
    }
#pragma unroll
    for (int v = 0; v < XV4; ++v) {
      if (slot_ok(v)) {
        float* __restrict__ d = xs + xz[v] * S + xo[v];
        if constexpr (kVecLds) {
          *reinterpret_cast<float4*>(d) = xreg[v];
        } else {
          d[0] = xreg[v].x; d[1] = xreg[v].y; d[2] = xreg[v].z; d[3] = xreg[v].w;
        }
      }
    }
  };

  f32x16n acc0 = {0}, acc1 = {0};
  const bool any_stage = q < ch.instr_end;
  int64_t g = bx;
  int64_t zbase = (g * 4 + wv) * NZT;  // first atom of this wavefront in the current group
  bool have = true;
  // one stage per group (one instruction, one type, K <= 64 -- every backward launch and linear_1): the weight slab is
  // loop invariant, later groups re-stage only their wavefront-private x slab and need no workgroup barrier at all
  const bool single_stage = any_stage && ch.instr_end - ch.instr_begin == 1 && a.n_types == 1 && ins.mul_in <= kNLK;
  if (any_stage) {
    if (!(a.dbg & 1)) load_stage(ins, t, k0, zbase, true);
    store_stage(0, true);
  }
  __syncthreads();
  int buf = 0;
  while (have) {
    const int64_t z = zbase + zl;
    const bool col_ok = (zlr < NZT) && (z < a.N);
    const int tzj = (a.types != nullptr && col_ok) ? (int)a.types[z] : 0;
    const int cur_t = t;
    // what comes next: the following stage of this group, else the first stage of this workgroup's next group
    bool last_of_group = !any_stage || !advance();
    int64_t zbase_next = zbase;
    bool next_valid = !last_of_group;
    if (last_of_group) {
      const int64_t gn = g + nblk;
      if (gn < ngroups) {
        g = gn;
        zbase_next = (gn * 4 + wv) * NZT;
        q = ch.instr_begin; t = 0; k0 = 0;
        if (any_stage) ins = a.instr[q];
        next_valid = true;
      }
    }
    if (next_valid && any_stage && !(a.dbg & 1)) load_stage(ins, t, k0, zbase_next, !single_stage);  // lands behind the MFMAs
    if (any_stage) {
      const bool bsel = col_ok && (a.n_types == 1 || tzj == cur_t);
      const float* __restrict__ wsb = ws + buf * (kNLK * kNLW) + j;
      const float* __restrict__ xb = xs + zl * S + m;
      // LDS operand reads of the next register batch are issued between the MFMAs of the current one
      constexpr int TB = 4;
      float bq[2][TB], a0[2][TB], a1[2][TB];
#pragma unroll
      for (int i = 0; i < TB; ++i) {
        const int u = 2 * i + half;
        bq[0][i] = xb[u * D];
        a0[0][i] = wsb[u * kNLW];
        a1[0][i] = wsb[u * kNLW + 32];
      }
#pragma unroll
      for (int b = 0; b < ((a.dbg & 2) ? 1 : kNLK / 2 / TB); ++b) {
        if (b + 1 < kNLK / 2 / TB) {
#pragma unroll
          for (int i = 0; i < TB; ++i) {
            const int u = 2 * ((b + 1) * TB + i) + half;
            bq[(b + 1) & 1][i] = xb[u * D];
            a0[(b + 1) & 1][i] = wsb[u * kNLW];
            a1[(b + 1) & 1][i] = wsb[u * kNLW + 32];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TB; ++i) {
          const float bv = bsel ? bq[b & 1][i] : 0.f;
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[b & 1][i], bv, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[b & 1][i], bv, acc1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (last_of_group) {
      // result tile -> this wavefront's slab as [atom][w*d + m] (same padded stride), then contiguous runs per atom
      // (the x slab is free: all MFMAs of the group are issued; same wavefront, LDS operations complete in order)
      if (zlr < NZT && (!(a.dbg & 8) || acc0[0] == 12345.f)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int wl = (r & 3) + 8 * (r >> 2) + 4 * half;
          xs[zl * S + wl * D + m] = acc0[r];
          xs[zl * S + (wl + 32) * D + m] = acc1[r];
        }
      }
      acc0 = (f32x16n){0};
      acc1 = (f32x16n){0};
      const int run = cw * D;  // valid floats per atom
      const bool oal = ((a.dout | ch.o_off | (ch.c0 * D)) & 3) == 0;
      if (oal && cw == kNLW && zbase + NZT <= a.N && kVecLds) {
        // common case (wave-uniform): full aligned runs of a complete atom group
        const int64_t ob = zbase * a.dout + ch.o_off + (int64_t)ch.c0 * D;
#pragma unroll
        for (int v = 0; v < XV4; ++v) {
          if (slot_ok(v)) {
            float4 r = *reinterpret_cast<const float4*>(xs + xz[v] * S + xo[v]);
            const int64_t o = ob + (int64_t)xz[v] * a.dout + xo[v];
            r.x *= a.scale; r.y *= a.scale; r.z *= a.scale; r.w *= a.scale;
            if (a.addend != nullptr) {
              const float4 ad = *reinterpret_cast<const float4*>(a.addend + o);
              r.x += ad.x; r.y += ad.y; r.z += ad.z; r.w += ad.w;
            }
            if (!(a.dbg & 4) || r.x == 12345.f) *reinterpret_cast<float4*>(a.out + o) = r;
          }
        }
      } else {
#pragma unroll
        for (int v = 0; v < XV4; ++v) {
          const int64_t zg = zbase + xz[v];
          if (xz[v] < NZT && zg < a.N && xo[v] < run) {
            const float* __restrict__ sp = xs + xz[v] * S + xo[v];
            float4 r;
            if constexpr (kVecLds) {
              r = *reinterpret_cast<const float4*>(sp);
            } else {
              r = make_float4(sp[0], sp[1], sp[2], sp[3]);
            }
            const int64_t o = zg * a.dout + ch.o_off + (int64_t)ch.c0 * D + xo[v];
            r.x *= a.scale; r.y *= a.scale; r.z *= a.scale; r.w *= a.scale;
            if (oal && xo[v] + 3 < run) {
              if (a.addend != nullptr) {
                const float4 ad = *reinterpret_cast<const float4*>(a.addend + o);
                r.x += ad.x; r.y += ad.y; r.z += ad.z; r.w += ad.w;
              }
              *reinterpret_cast<float4*>(a.out + o) = r;
            } else {
              const float rv[4] = {r.x, r.y, r.z, r.w};
              for (int e = 0; e < 4; ++e)
                if (xo[v] + e < run) a.out[o + e] = rv[e] + (a.addend != nullptr ? a.addend[o + e] : 0.f);
            }
          }
        }
      }
    }
    if (next_valid && any_stage) {
      if (single_stage) {
        store_stage(buf, false);  // wavefront-private slab: in-order LDS, no barrier
      } else {
        // LDS-only barriers (s_waitcnt lgkmcnt(0); s_barrier): __syncthreads() would also drain vmcnt, i.e. wait for
        // the result stores just issued (CDNA4 counts stores in vmcnt) at every group boundary
        if (!(a.dbg & 16)) nl_lds_barrier();  // all wavefronts are done with ws[buf ^ 1] and with their own slab
        if (!(a.dbg & 32)) store_stage(buf ^ 1, true);
        if (!(a.dbg & 16)) nl_lds_barrier();
        buf ^= 1;
      }
    }
    zbase = zbase_next;
    have = next_valid;
  }
}

__global__ __launch_bounds__(256, 2) void node_linear_mfma_kernel(const NodeLinearArgs<float> a) {
  __shared__ __align__(16) float ws[2 * kNLK * kNLW];  // 32 KiB: double-buffered weight slab
  __shared__ __align__(16) float xs[4 * kNLXS];        // 4 x 8.1 KiB: per-wavefront x / result slabs
  // exact 1-D grid: chunk c owns workgroups [blk_begin[c], blk_begin[c+1])
  int c = 0;
  while (c + 1 < a.n_chunks && (int)blockIdx.x >= a.blk_begin[c + 1]) ++c;
  const NodeChunk ch = a.chunks[c];
  const int bx = (int)blockIdx.x - a.blk_begin[c];
  const int nblk = a.blk_begin[c + 1] - a.blk_begin[c];
  switch (ch.d) {
    case 1: node_linear_mfma_block<1>(a, ch, bx, nblk, ws, xs); break;
    case 3: node_linear_mfma_block<3>(a, ch, bx, nblk, ws, xs); break;
    case 5: node_linear_mfma_block<5>(a, ch, bx, nblk, ws, xs); break;
    case 7: node_linear_mfma_block<7>(a, ch, bx, nblk, ws, xs); break;
    case 9: node_linear_mfma_block<9>(a, ch, bx, nblk, ws, xs); break;
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
      const int rc = nqa_node_linear(dtype, x, weights, addend, out, atom_types,
                                     static_cast<const NodeChunk*>(chunk_table) + c0, nc, instr_table, n_instr, n_types,
                                     weight_stride, dim_in, dim_out, num_nodes, scale, chunk_width, stream);
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
      // atom groups per workgroup: enough workgroups to fill the chip a few times over, few enough that each
      // amortises its first (exposed) operand fetch over several groups
      int64_t total_groups = 0;
      for (int c = 0; c < n_chunks; ++c) {
        const int d = a.chunks[c].d;
        if (d != 1 && d != 3 && d != 5 && d != 7 && d != 9) {
          set_error("nqa_node_linear: irrep dimension above 9 (l > 4)");
          return NQA_ERR_UNSUPPORTED;
        }
        const int per_grp = 4 * (32 / d);
        total_groups += (num_nodes + per_grp - 1) / per_grp;
      }
      static const int gpb_env = [] {
        const char* e = std::getenv("NQA_NODE_GPB");
        return e ? std::atoi(e) : 0;
      }();
      int64_t gpb = gpb_env > 0 ? gpb_env : (total_groups + 1023) / 1024;
      if (gpb < 1) gpb = 1;
      if (gpb > 8) gpb = 8;
      int64_t nblk = 0;
      for (int c = 0; c < n_chunks; ++c) {
        a.blk_begin[c] = (int32_t)nblk;
        const int per_grp = 4 * (32 / a.chunks[c].d);
        const int64_t groups = (num_nodes + per_grp - 1) / per_grp;
        nblk += (groups + gpb - 1) / gpb;
      }
      a.blk_begin[n_chunks] = (int32_t)nblk;
      if (nblk == 0) return NQA_OK;
      const dim3 mgrid((unsigned)nblk);
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
