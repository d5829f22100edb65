// Node-side chain of one layer boundary in ONE launch (inference):
//
//   forward   h  = linear_2(a) + sc_prev          x' = gate(h)            y = scale * linear_1'(x')    s = sc'(x', type)
//   backward  g' = scale * linear_1'^T(g_y) + sc'^T(g_s, type)   g_h = gate'(g', h)   g_a = linear_2^T(g_h)
//
// i.e. what the reference runs as e3nn `o3.Linear` -> `+ sc` -> `Gate` -> {`FullyConnectedTensorProduct`, `o3.Linear`}
// across the end of one InteractionBlock, the ConvNetLayer's nonlinearity and the start of the next InteractionBlock
// (nequip/nn/interaction_block.py:175-177,201-204; nequip/nn/convnetlayer.py:156-170), and what this package ran as
// four launches each way (nqa_node_linear x3 + nqa_gate): 21 launches, 0.6 ms of the 3.1 ms cfg-3 step, each of them
// latency-bound on 10 125 rows.
//
// A workgroup owns G = 16 atoms for the whole chain.  The launch is a list of *phases*; a phase is a list of 64-channel
// output chunks (of possibly several destination tensors); a chunk sums over *instructions* (input block -> output
// block matrices) that may read different source tensors with different weight sets:
//   phase 1 writes its rows to HBM (h is needed by the backward, g' by the gate derivative), then `__threadfence()` +
//   barrier, and phase 2 reads those very rows back -- from L2 -- through a *gate view*: the forward / backward gate
//   formulas are applied while the operand slab is staged into LDS, so the gated tensor is never materialised and the
//   gate costs no launch and no HBM round trip.
// GEMM tiling (exact fp32, v_mfma_f32_32x32x2_f32 -- same arithmetic as nqa_node_linear): D[w][col] with w = 64 output
// channels (two row tiles) and col = (atom, m) flattened over the group's atoms (G * d columns, ceil(G d / 32) column
// tiles); the (row, column) tiles of a chunk are dealt to the 8 wavefronts.  Per stage (instruction, 64-channel K slab,
// atom type) the weight slab [64][64] and the operand slab [G][64 d] are staged through double-buffered LDS, the next
// stage's global loads are issued before the current stage's MFMAs.  Typed weights (self-connection): the operand slab
// is staged once per K slab, the columns of atoms of another type are zeroed in the B operand.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>

#include "plan.h"

namespace nqa {

using f32x16c = __attribute__((ext_vector_type(16))) float;

constexpr int kCW = 64;    // output channels per chunk
constexpr int kCK = 64;    // input channels per stage
constexpr int kCT = 512;   // threads per workgroup (8 wavefronts)
constexpr int kCWV = kCT / 64;
constexpr int kMaxT = 2;   // tiles per wavefront (2 * ceil(G d / 32) <= 16 tiles)

struct ChainChunk {
  int32_t dst, o_off, d, mul_out;
  int32_t c0, instr_begin, instr_end, flags;  // flags bit 0: add the destination's addend
};
struct ChainInstr {
  int32_t src, x_off, mul_in, wset;
  int32_t w_off, kind, act, store_off;  // store_off >= 0: also write the staged values to the source's `store` rows
  float scale, cst;                     // scale multiplies the staged operand; cst = normalize2mom constant of `act`
  int32_t aux0, aux1, aux2, len;        // see ChainKind
  int32_t pad0, pad1;
};
static_assert(sizeof(ChainChunk) == 32 && sizeof(ChainInstr) == 64, "table layout is part of the C ABI");

// What an instruction's operand block is (u = channel within the block, m = component, d = the chunk's irrep dimension):
//   0 plain        rows[x_off + u d + m]
//   1 gate fwd, scalar   act(rows[x_off + u])                                        (d = 1)
//   2 gate fwd, gated    act(rows[aux0 + u]) * rows[x_off + u d + m]                 aux0 = column of the gate scalars
//   3 gate bwd, scalar   rows[aux1 + u] * act'(gin[x_off + u])                       rows = grad w.r.t. gate output,
//   4 gate bwd, gated    act(gin[aux0 + u]) * rows[aux1 + u d + m]                   gin = gate input; x_off = column
//   5 gate bwd, gate     act'(gin[x_off + u]) * sum_{m < len} rows[aux1 + u len + m] * gin[aux2 + u len + m]   (d = 1)
//                                                                                     of the gate input the value belongs to
enum ChainKind { kPlain = 0, kFwdScalar = 1, kFwdGated = 2, kBwdScalar = 3, kBwdGated = 4, kBwdGate = 5 };

struct ChainSrc {
  const float* p0;       // the rows (kinds 0-2: the tensor itself / the gate input; kinds 3-5: grad w.r.t. the gate output)
  const float* p1;       // kinds 3-5: the gate input
  float* store;          // where instructions with store_off >= 0 write what they stage, [N, dim_store]
  int32_t dim0, dim1, dim_store, pad;
};
struct ChainDst {
  float* p;
  const float* addend;
  int32_t dim;
  float scale;
};
struct ChainW {
  const float* p;
  int64_t wstride;
  int32_t n_types, pad;
};
constexpr int kChainSlots = 3;  // sources / destinations / weight sets per launch
struct ChainArgs {
  ChainSrc src[kChainSlots];
  ChainDst dst[kChainSlots];
  ChainW w[kChainSlots];
  const ChainChunk* chunks;
  const ChainInstr* instr;
  const int64_t* types;
  int64_t N;
  int32_t phase_begin[4];  // chunk ranges: phase p = [phase_begin[p], phase_begin[p + 1])
  int32_t n_phases;
  int32_t dmax;
  int32_t dbg;  // ablation switches (NQA_CHAIN_DBG; timing probes, wrong results), 0 in production
};

// Descriptor selection without dynamic indexing of the kernel-argument block (a dynamically indexed by-value argument is
// copied to scratch memory; the ids are wave-uniform, so these are scalar selects)
__device__ __forceinline__ ChainSrc chain_pick(const ChainSrc (&v)[kChainSlots], int i) {
  return i == 0 ? v[0] : (i == 1 ? v[1] : v[2]);
}
__device__ __forceinline__ ChainDst chain_pick(const ChainDst (&v)[kChainSlots], int i) {
  return i == 0 ? v[0] : (i == 1 ? v[1] : v[2]);
}
__device__ __forceinline__ ChainW chain_pick(const ChainW (&v)[kChainSlots], int i) {
  return i == 0 ? v[0] : (i == 1 ? v[1] : v[2]);
}

__device__ __forceinline__ float chain_act(int act, float x, float cst) {
  if (act == 1) return cst * x / (1.0f + __expf(-x));
  if (act == 2) return cst * tanhf(x);
  return x;
}
__device__ __forceinline__ float chain_act_grad(int act, float x, float cst) {
  if (act == 1) {
    const float s = 1.0f / (1.0f + __expf(-x));
    return cst * s * (1.0f + x * (1.0f - s));
  }
  if (act == 2) {
    const float t = tanhf(x);
    return cst * (1.0f - t * t);
  }
  return 1.0f;
}

template <int D, int kCG>
__device__ __forceinline__ void chain_chunk(const ChainArgs& a, const ChainChunk& ch, int64_t z0, float* __restrict__ ws,
                                            float* __restrict__ xs) {
  constexpr int P = D == 1 ? 4 : ((D + 3) / 4) * 4;
  constexpr int S = kCK * D + P;                 // operand-slab stride per atom (floats)
  constexpr int XBUF = kCG * S;                  // floats per operand buffer
  constexpr int NCOL = kCG * D;
  constexpr int NCT = (NCOL + 31) / 32;
  constexpr int NT = 2 * NCT;                    // (row tile, column tile) pairs of a chunk
  constexpr int TPW = (NT + kCWV - 1) / kCWV;    // tiles per wavefront
  static_assert(TPW <= kMaxT, "too many tiles per wavefront");
  constexpr int XE = (kCG * kCK * D + kCT - 1) / kCT;  // operand elements per thread per stage
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int cw = min(kCW, ch.mul_out - ch.c0);

  // this wavefront's tiles: t = wv + 8 i  ->  row tile t & 1, column tile t >> 1
  int xoff[TPW];     // LDS offset of this lane's column (atom * S + m), or -1
  int zt[TPW];       // this lane's atom within the group (for the type mask)
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int t = wv + kCWV * i;
    const int col = (t >> 1) * 32 + l31;
    const bool ok = t < NT && col < NCOL;
    const int z = ok ? col / D : 0;
    zt[i] = z;
    xoff[i] = ok ? z * S + (col - z * D) : -1;
  }
  int tz[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i) tz[i] = (a.types != nullptr && z0 + zt[i] < a.N) ? (int)a.types[z0 + zt[i]] : 0;

  f32x16c acc[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i) acc[i] = (f32x16c){0};

  // ---- stages: (instruction q, K slab k0, atom type t), t innermost.  The global loads of the next stage are issued
  //      before the MFMAs of the current one and land in registers; the gate arithmetic is applied when they move to the
  //      other LDS buffer after the MFMAs.  (A second register set -- loads two stages ahead -- was tried: it does not fit
  //      the register file next to the accumulators, 2050 spilled registers at 40 atoms per workgroup.)  The operand slab is
  //      re-staged for every atom type of a typed stage (same rows again, from L1 / L2) so that the weight and operand
  //      buffers advance in lock step.
  struct Stage {
    ChainInstr ins;
    int q, k0, t;
    bool valid;
  };
  struct Regs {
    float4 w[2];
    float x[XE], y[XE];
  };
  auto ntypes = [&](const ChainInstr& i) { return chain_pick(a.w, i.wset).n_types; };
  auto next_stage = [&](const Stage& c) {
    Stage n = c;
    if (!c.valid) return n;
    n.t = c.t + 1;
    if (n.t >= ntypes(c.ins)) {
      n.t = 0;
      n.k0 = c.k0 + kCK;
      if (n.k0 >= c.ins.mul_in) {
        n.k0 = 0;
        n.q = c.q + 1;
        n.valid = n.q < ch.instr_end;
        if (n.valid) n.ins = a.instr[n.q];
      }
    }
    return n;
  };
  auto load_stage = [&](const Stage& st, Regs& R) {
    const ChainInstr& si = st.ins;
    {  // weight slab rows k0 .. k0 + 63, columns c0 .. c0 + 63 of W_t [mul_in][mul_out]
      const ChainW W = chain_pick(a.w, si.wset);
      const float* __restrict__ wb = W.p + (int64_t)st.t * W.wstride + si.w_off + ch.c0;
      const bool al = ((ch.mul_out | ch.c0 | si.w_off) & 3) == 0 && (W.wstride & 3) == 0;
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        const int idx = tid + v * kCT;  // float4 index in the [64 k][16] slab
        const int k = idx >> 4, c4 = (idx & 15) * 4;
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (st.k0 + k < si.mul_in) {
          const float* __restrict__ p = wb + (int64_t)(st.k0 + k) * ch.mul_out + c4;
          if (al && c4 + 3 < cw) {
            r = *reinterpret_cast<const float4*>(p);
          } else {
            if (c4 + 0 < cw) r.x = p[0];
            if (c4 + 1 < cw) r.y = p[1];
            if (c4 + 2 < cw) r.z = p[2];
            if (c4 + 3 < cw) r.w = p[3];
          }
        }
        R.w[v] = r;
      }
    }
    // operand slab: raw loads only (kBwdGate sums 2 len products per element: evaluated here, d = 1 blocks only)
    const ChainSrc sr = chain_pick(a.src, si.src);
    const int sk0 = st.k0;
    const int kk = min(kCK, si.mul_in - sk0) * D;  // valid floats per atom in this slab
    // (the kind is wave-uniform: one loop per kind instead of a switch per element)
    auto elem = [&](int e, int64_t& zz, int& rr, int& u) {
      const int idx = tid + e * kCT;
      const int z = idx / (kCK * D), r = idx - z * (kCK * D);
      const bool ok = idx < kCG * kCK * D && r < kk && z0 + z < a.N;
      zz = ok ? z0 + z : z0;                      // clamped row: unpredicated loads, masked when stored to LDS
      rr = ok ? r : 0;
      u = sk0 + rr / D;                           // channel within the instruction's block
    };
    const int xb = si.x_off + sk0 * D;
    if (si.kind == kPlain) {
#pragma unroll
      for (int e = 0; e < XE; ++e) {
        int64_t zz; int rr, u; elem(e, zz, rr, u);
        R.x[e] = sr.p0[zz * sr.dim0 + xb + rr];
      }
    } else if (si.kind == kFwdScalar) {
#pragma unroll
      for (int e = 0; e < XE; ++e) {
        int64_t zz; int rr, u; elem(e, zz, rr, u);
        R.x[e] = sr.p0[zz * sr.dim0 + si.x_off + u];
      }
    } else if (si.kind == kFwdGated) {
#pragma unroll
      for (int e = 0; e < XE; ++e) {
        int64_t zz; int rr, u; elem(e, zz, rr, u);
        R.x[e] = sr.p0[zz * sr.dim0 + xb + rr];
        R.y[e] = sr.p0[zz * sr.dim0 + si.aux0 + u];
      }
    } else if (si.kind == kBwdScalar) {
#pragma unroll
      for (int e = 0; e < XE; ++e) {
        int64_t zz; int rr, u; elem(e, zz, rr, u);
        R.x[e] = sr.p0[zz * sr.dim0 + si.aux1 + u];
        R.y[e] = sr.p1[zz * sr.dim1 + si.x_off + u];
      }
    } else if (si.kind == kBwdGated) {
#pragma unroll
      for (int e = 0; e < XE; ++e) {
        int64_t zz; int rr, u; elem(e, zz, rr, u);
        R.x[e] = sr.p0[zz * sr.dim0 + si.aux1 + sk0 * D + rr];
        R.y[e] = sr.p1[zz * sr.dim1 + si.aux0 + u];
      }
    } else if (si.kind == kBwdGate) {
#pragma unroll
      for (int e = 0; e < XE; ++e) {
        int64_t zz; int rr, u; elem(e, zz, rr, u);
        const float* __restrict__ row0 = sr.p0 + zz * sr.dim0 + si.aux1 + u * si.len;
        const float* __restrict__ row1 = sr.p1 + zz * sr.dim1 + si.aux2 + u * si.len;
        float acc2 = 0.f;
#pragma unroll
        for (int m = 0; m < 9; ++m) {
          const int mm = m < si.len ? m : 0;
          const float v = row0[mm] * row1[mm];
          acc2 += m < si.len ? v : 0.f;
        }
        R.x[e] = acc2;
        R.y[e] = sr.p1[zz * sr.dim1 + si.x_off + u];
      }
    }
  };
  auto store_stage = [&](const Stage& st, const Regs& R, int buf) {
    const ChainInstr& xins = st.ins;
#pragma unroll
    for (int v = 0; v < 2; ++v) *reinterpret_cast<float4*>(ws + buf * (kCK * kCW) + (tid + v * kCT) * 4) = R.w[v];
    const ChainSrc sr = chain_pick(a.src, xins.src);
    const int kk = min(kCK, xins.mul_in - st.k0) * D;
#pragma unroll
    for (int e = 0; e < XE; ++e) {
      const int idx = tid + e * kCT;
      if (idx < kCG * kCK * D) {
        const int z = idx / (kCK * D), r = idx - z * (kCK * D);
        const bool ok = r < kk && z0 + z < a.N;
        float v = R.x[e];
        switch (xins.kind) {
          case kFwdScalar: v = chain_act(xins.act, v, xins.cst); break;
          case kFwdGated: v = chain_act(xins.act, R.y[e], xins.cst) * v; break;
          case kBwdScalar: v = v * chain_act_grad(xins.act, R.y[e], xins.cst); break;
          case kBwdGated: v = chain_act(xins.act, R.y[e], xins.cst) * v; break;
          case kBwdGate: v = v * chain_act_grad(xins.act, R.y[e], xins.cst); break;
          default: break;
        }
        v = ok ? v : 0.f;
        if (ok && xins.store_off >= 0 && st.t == 0)
          sr.store[(z0 + z) * sr.dim_store + xins.store_off + st.k0 * D + r] = v;
        xs[buf * XBUF + z * S + r] = v * xins.scale;
      }
    }
  };
  auto compute = [&](const Stage& st, int buf) {
    const bool typed = ntypes(st.ins) > 1;
    const float* __restrict__ wsb = ws + buf * (kCK * kCW);
    const float* __restrict__ xsb = xs + buf * XBUF;
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      const int tt = wv + kCWV * i;
      if (tt < NT) {  // (wave-uniform)
        const int rt = tt & 1;
        // columns that do not exist, or belong to an atom of another type, contribute zero: a 0 / 1 factor on unconditional
        // LDS reads (a predicated read compiles into a branch + wait in front of every MFMA)
        const float mf = (xoff[i] >= 0 && (!typed || tz[i] == st.t)) ? 1.f : 0.f;
        const float* __restrict__ ap = wsb + rt * 32 + l31 + half * kCW;
        const float* __restrict__ bp = xsb + (xoff[i] >= 0 ? xoff[i] : 0) + half * D;
        constexpr int KB = 8;  // operands of 8 MFMAs are requested together, the next batch behind the current MFMAs
        float av[2][KB], bv[2][KB];
#pragma unroll
        for (int j = 0; j < KB; ++j) {
          av[0][j] = ap[(2 * j) * kCW];
          bv[0][j] = bp[(2 * j) * D];
        }
#pragma unroll
        for (int b = 0; b < kCK / 2 / KB; ++b) {
          if (b + 1 < kCK / 2 / KB) {
#pragma unroll
            for (int j = 0; j < KB; ++j) {
              av[(b + 1) & 1][j] = ap[(2 * ((b + 1) * KB + j)) * kCW];
              bv[(b + 1) & 1][j] = bp[(2 * ((b + 1) * KB + j)) * D];
            }
          }
#pragma unroll
          for (int j = 0; j < KB; ++j)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[b & 1][j], bv[b & 1][j] * mf, acc[i], 0, 0, 0);
        }
      }
    }
  };

  Stage cur, nxt;
  nxt.q = ch.instr_begin;
  nxt.k0 = 0;
  nxt.t = 0;
  nxt.valid = nxt.q < ch.instr_end;
  nxt.ins = nxt.valid ? a.instr[nxt.q] : ChainInstr{};
  cur = nxt;
  cur.valid = false;
  Regs R = {};
  int buf = 1;
  // (one call site per helper: the first pass through the loop only stages, the last one only computes)
  while (true) {
    if (nxt.valid && !(a.dbg & 1)) load_stage(nxt, R);        // in flight behind the MFMAs of `cur`
    if (cur.valid && !(a.dbg & 2)) compute(cur, buf);
    if (nxt.valid && !(a.dbg & 4)) store_stage(nxt, R, buf ^ 1);
    __syncthreads();
    if (!nxt.valid) break;
    cur = nxt;
    buf ^= 1;
    nxt = next_stage(cur);
  }

  // ---- epilogue: tiles -> operand slab 0 as [atom][w * d + m], then contiguous runs per atom to the destination ------
  if (a.dbg & 8) return;
  float* __restrict__ slab = xs;
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int tt = wv + kCWV * i;
    if (tt < NT && xoff[i] >= 0) {
      const int rt = tt & 1;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int wl = 32 * rt + (r & 3) + 8 * (r >> 2) + 4 * half;
        slab[xoff[i] + wl * D] = acc[i][r];
      }
    }
  }
  __syncthreads();
  {
    const ChainDst ds = chain_pick(a.dst, ch.dst);
    const int run = cw * D;
    const bool add = (ch.flags & 1) && ds.addend != nullptr;
#pragma unroll
    for (int e = 0; e < XE; ++e) {
      const int idx = tid + e * kCT;
      const int z = idx / (kCK * D), r = idx - z * (kCK * D);
      if (idx < kCG * kCK * D && r < run && z0 + z < a.N) {
        const int64_t o = (z0 + z) * ds.dim + ch.o_off + (int64_t)ch.c0 * D + r;
        float v = ds.scale * slab[z * S + r];
        if (add) v += ds.addend[o];
        ds.p[o] = v;
      }
    }
  }
  __syncthreads();
}

// One kernel per largest irrep dimension of the launch: the register allocation of a kernel is that of its widest
// chunk variant (92 / 125 / 153 / 179 / 232 VGPRs for d = 1 / 3 / 5 / 7 / 9), so a model without l = 4 blocks does not pay
// for them.
template <int DMAX, int kCG>
__global__ __launch_bounds__(kCT) void node_chain_kernel(const ChainArgs a) {
  extern __shared__ __align__(16) unsigned char nqa_chain_smem[];
  float* ws = reinterpret_cast<float*>(nqa_chain_smem);  // 2 x [64][64]
  float* xs = ws + 2 * kCK * kCW;                         // 2 x [G][64 dmax + pad]
  const int64_t z0 = (int64_t)blockIdx.x * kCG;
  for (int p = 0; p < a.n_phases; ++p) {
    for (int c = a.phase_begin[p]; c < a.phase_begin[p + 1]; ++c) {
      const ChainChunk ch = a.chunks[c];
      if (ch.d == 1) chain_chunk<1, kCG>(a, ch, z0, ws, xs);
      if constexpr (DMAX >= 3) { if (ch.d == 3) chain_chunk<3, kCG>(a, ch, z0, ws, xs); }
      if constexpr (DMAX >= 5) { if (ch.d == 5) chain_chunk<5, kCG>(a, ch, z0, ws, xs); }
      if constexpr (DMAX >= 7) { if (ch.d == 7) chain_chunk<7, kCG>(a, ch, z0, ws, xs); }
      if constexpr (DMAX >= 9) { if (ch.d == 9) chain_chunk<9, kCG>(a, ch, z0, ws, xs); }
    }
    if (p + 1 < a.n_phases) {
      // rows written by this workgroup in phase p are read back (through the gate views) in phase p + 1
      __threadfence();
      __syncthreads();
    }
  }
}

}  // namespace nqa

using namespace nqa;

extern "C" {

// Host-side mirror of the argument block (plain pointers and sizes; see include/nequip_amd.h)
int nqa_node_chain(const nqa_chain_desc* desc, nqa_stream stream) {
  if (desc == nullptr || desc->num_nodes < 0 || desc->n_phases < 1 || desc->n_phases > 3 || desc->chunks == nullptr ||
      desc->instr == nullptr) {
    set_error("nqa_node_chain: invalid descriptor");
    return NQA_ERR_INVALID;
  }
  if (desc->num_nodes == 0) return NQA_OK;
  ChainArgs a{};
  int dmax = 1;
  for (int i = 0; i < kChainSlots; ++i) {
    a.src[i].p0 = static_cast<const float*>(desc->src[i].rows);
    a.src[i].p1 = static_cast<const float*>(desc->src[i].gate_input);
    a.src[i].store = static_cast<float*>(desc->src[i].store);
    a.src[i].dim0 = desc->src[i].dim;
    a.src[i].dim1 = desc->src[i].gate_input_dim;
    a.src[i].dim_store = desc->src[i].store_dim;
    if (a.src[i].p1 == nullptr) {  // (clamped, never-used loads of the second row need a valid address)
      a.src[i].p1 = a.src[i].p0;
      a.src[i].dim1 = a.src[i].dim0;
    }
    a.dst[i].p = static_cast<float*>(desc->dst[i].rows);
    a.dst[i].addend = static_cast<const float*>(desc->dst[i].addend);
    a.dst[i].dim = desc->dst[i].dim;
    a.dst[i].scale = (float)desc->dst[i].scale;
    a.w[i].p = static_cast<const float*>(desc->weights[i].data);
    a.w[i].wstride = desc->weights[i].stride;
    a.w[i].n_types = desc->weights[i].n_types < 1 ? 1 : desc->weights[i].n_types;
    if (a.w[i].n_types > 1 && desc->atom_types == nullptr) {
      set_error("nqa_node_chain: typed weights need atom_types");
      return NQA_ERR_INVALID;
    }
  }
  a.chunks = static_cast<const ChainChunk*>(desc->chunks);
  a.instr = static_cast<const ChainInstr*>(desc->instr);
  a.types = desc->atom_types;
  a.N = desc->num_nodes;
  a.n_phases = desc->n_phases;
  for (int p = 0; p <= desc->n_phases; ++p) a.phase_begin[p] = desc->phase_begin[p];
  dmax = desc->max_irrep_dim;
  if (dmax != 1 && dmax != 3 && dmax != 5 && dmax != 7 && dmax != 9) {
    set_error("nqa_node_chain: max_irrep_dim must be one of 1, 3, 5, 7, 9 (l <= 4)");
    return NQA_ERR_UNSUPPORTED;
  }
  a.dmax = dmax;
  {
    static const int dbg = [] {
      const char* e = std::getenv("NQA_CHAIN_DBG");
      return e ? std::atoi(e) : 0;
    }();
    a.dbg = dbg;
  }
  // Atoms per workgroup: the group is as large as the LDS slabs allow when that lets the whole launch run as one wave of
  // workgroups (a stage's MFMA time then covers the latency of the next stage's loads), else 16.
  static const int num_cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
  }();
  // (40 atoms per workgroup -- one wave of workgroups for the cfg-3 box -- was measured too: 1.69 vs 1.55 ms per step for
  // the four launches, its 25 staged elements per thread spill)
  const int G = 16;
  (void)num_cus;
  const int P = dmax == 1 ? 4 : ((dmax + 3) / 4) * 4;
  const size_t smem = (size_t)(2 * kCK * kCW + 2 * G * (kCK * dmax + P)) * sizeof(float);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t nblk = (a.N + G - 1) / G;
  const dim3 grid((unsigned)nblk), blk(kCT);
  hipError_t lerr = hipSuccess;
#define NQA_CHAIN_LAUNCH(DM, GG)                                                                                      \
  {                                                                                                                   \
    if (smem > 64 * 1024) {                                                                                           \
      static bool configured = false;                                                                                 \
      if (!configured) {                                                                                              \
        lerr = hipFuncSetAttribute(reinterpret_cast<const void*>(node_chain_kernel<DM, GG>),                          \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                           \
        configured = lerr == hipSuccess;                                                                              \
      }                                                                                                               \
    }                                                                                                                 \
    if (lerr == hipSuccess) hipLaunchKernelGGL((node_chain_kernel<DM, GG>), grid, blk, smem, s, a);                   \
  }
  switch (dmax) {
    case 1: NQA_CHAIN_LAUNCH(1, 16) break;
    case 3: NQA_CHAIN_LAUNCH(3, 16) break;
    case 5: NQA_CHAIN_LAUNCH(5, 16) break;
    case 7: NQA_CHAIN_LAUNCH(7, 16) break;
    default: NQA_CHAIN_LAUNCH(9, 16) break;
  }
#undef NQA_CHAIN_LAUNCH
  if (lerr != hipSuccess) {
    set_error("nqa_node_chain: cannot reserve the LDS slabs");
    return NQA_ERR_LAUNCH;
  }
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_error(std::string("nqa_node_chain: ") + hipGetErrorString(err));
    return NQA_ERR_LAUNCH;
  }
  return NQA_OK;
}

}  // extern "C"
