// node_fused.h -- the node side of a layer boundary as ONE launch per direction (included at the end of node_ops.hip).
//
// Between two tensor products the reference runs, on the N atom rows (nequip/nn/convnetlayer.py:156-170,
// nequip/nn/interaction_block.py:175-177,201-204):
//     h  = linear_2(y) + sc                       (layer L;   h = scalars (+) gates (+) gated, "pre-gate" rows)
//     x  = Gate(h)                                (layer L)
//     sc' = sc_{L+1}(x, node_attrs);   x1 = linear_1_{L+1}(x) / sqrt(avg_num_neighbors)      (layer L+1)
// i.e. gate, linear_1 and the self-connection are three launches reading / writing [N, D] rows, and four more in the
// backward pass.  Here `Gate` never materialises:
//   * forward:  ONE launch reads h, applies the gate while the operand slab is staged (act(scalars) when the slab goes to
//     LDS; act(gate scalars) into a small LDS table that multiplies the B fragments of the gated blocks) and feeds BOTH
//     consumers -- linear_1 (untyped weights, scale 1/sqrt(avg)) and the per-type pre-contracted self-connection -- whose
//     output tiles go to two destinations;
//   * backward: ONE launch accumulates linear_1^T(g_x1) + sc^T(g_sc) (two operand sets, one accumulator tile per unit) and
//     applies the gate's backward in the epilogue (act'(scalars); for a gated block the values times act(gate) and the
//     gate scalars' gradient act'(gate) * sum_m g * v), writing the gradient of h directly.
// The GEMM core is the per-wavefront pipeline of node_linear_wave_bf16_unit<D, F16 = true> (two-plane fp16 split with a
// running per-column exponent, weights packed by nqa_node_weights_pack, fp32-accurate); what is new is around it:
// operand sets per instruction, destinations per chunk, the load-time transform and the epilogues.
namespace nqa {

constexpr int kNFMaxChunks = 24;
constexpr int kNFMaxInstr = 48;
constexpr int kNFGateStride = kNLK3 + 4;      // floats per atom row of the gate table in LDS (padded: rows 144 B apart)
constexpr int kNFGS = 10 * kNFGateStride + 24; // floats of gate table per wavefront (d = 3: 10 atoms; larger d: fewer)

struct NFSet {  // operand set: rows + packed weights
  const float* __restrict__ x;
  const nl_u32x4* __restrict__ wf;
  const int32_t* __restrict__ wexp;
  int64_t frag_stride;
  int32_t din, n_types, exp_stride, pad;
};
struct NFDst {
  float* __restrict__ out;
  const float* __restrict__ addend;
  int32_t dout;
  float scale;
};
struct NFInstr {
  int32_t x_off, mul_in, frag_off, exp_off;
  int32_t gate_off;  // >= 0: the block's gate scalars (B operand = value * act(gate)); -1: act on the values; -2: plain
  int32_t act;
  float cst;
  int32_t set;
};
static_assert(sizeof(NFInstr) == 32, "NFInstr");
struct NFChunk {
  int32_t o_off, d, mul_out, c0, instr_begin, instr_end;
  int32_t dst;
  int32_t ep;      // epilogue: 0 plain (scale, addend); 1 scalar block of a gate backward; 2 gated block of a gate backward
  int32_t ev_off;  // ep 1 / 2: offset of the block's values in h and in out (channel 0 of the block)
  int32_t eg_off;  // ep 2: offset of the block's gate scalars in h and in out
  int32_t act;
  float cst;
};
static_assert(sizeof(NFChunk) == 48, "NFChunk");

struct NodeFusedArgs {
  NFSet sets[2];
  NFDst dsts[2];
  const float* __restrict__ h;  // epilogue operand (pre-gate rows of the gate whose backward the epilogue applies)
  const int64_t* __restrict__ types;
  // optional: the order in which the units walk the atoms (row r of the launch is atom perm[r]).  ANY permutation gives the
  // same results; one that groups the atoms by type lets a unit skip the typed stages of the types it does not hold.
  const int32_t* __restrict__ perm;
  int64_t N;
  int32_t hdim, n_chunks, n_groups, pad;
  NFChunk chunks[kNFMaxChunks];
  NFInstr instr[kNFMaxInstr];
  int32_t grp_begin[kNFMaxChunks + 1];
  int32_t grp_chunk0[kNFMaxChunks];
  int32_t grp_n[kNFMaxChunks];
};
static_assert(sizeof(NodeFusedArgs) <= 3900, "kernel arguments must stay below 4 KiB");

// Activations of the fused stage: one v_exp_f32 + one v_rcp_f32 per value (1 ulp each) instead of expf() and an IEEE
// division -- a unit re-activates the gate scalars of every slab it stages (8 per lane and stage), which at ~20
// instructions per silu was a third of the stage's vector instructions.
__device__ __forceinline__ float nf_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float nf_act(int act, float x, float cst) {
  if (act == 1) return cst * x * nf_sigmoid(x);
  if (act == 2) return cst * tanhf(x);
  return x;
}
__device__ __forceinline__ float nf_act_grad(int act, float x, float cst) {
  if (act == 1) {
    const float s = nf_sigmoid(x);
    return cst * s * (1.f + x * (1.f - s));
  }
  if (act == 2) {
    const float t = tanhf(x);
    return cst * (1.f - t * t);
  }
  return 1.f;
}

struct NFStage {  // one (instruction, atom type, 32-channel K slab) step of a chunk; wave-uniform
  int q, t, k0;
  int mul_in, x_off, gate_off, act, set, n_types, din, frag_off, exp_off;
  float cst;
  const float* xp;
  bool valid;
};

// PIPE: the stage loop of node_linear_wave_bf16_unit<D, true, PIPE = true> (node_ops.hip): one wait per stage, at its top, for
// loads requested a full stage earlier -- the x slab, the gate scalars and BOTH K blocks' weight fragments of stage s + 1 go
// out right after the slab of stage s is in LDS (second fragment buffer, loop unrolled by two); two wavefronts per SIMD.
template <int D, bool PIPE>
__device__ __forceinline__ void node_fused_unit(const NodeFusedArgs& a, const NFChunk& ch, int64_t g,
                                                float* __restrict__ xs, float* __restrict__ gs) {
  constexpr int NZT = 32 / D;
  constexpr int P = D == 1 ? 1 : ((D + 3) / 4) * 4;
  constexpr int S = kNLK3 * D + P;
  constexpr bool kVecLds = (S % 4) == 0;
  constexpr int RUN4 = kNLK3 * D / 4;
  constexpr int XV4 = (NZT * RUN4 + 63) / 64;
  constexpr int GQ = kNLK3 / 4;                      // float4 of gate scalars per atom and stage
  constexpr int XG4 = (NZT * GQ + 63) / 64;          // ... per lane
  constexpr int GS = kNFGateStride;
  static_assert(NZT * S <= kNLXS, "slab too small");
  static_assert(D == 1 || NZT * GS <= kNFGS, "gate table too small");
  const int lane = threadIdx.x & 63;
  const int half = lane >> 5, j = lane & 31;
  const int zlr = j / D, m = j - zlr * D;
  const int zl = min(zlr, NZT - 1);
  const int cw = min(kNLW, ch.mul_out - ch.c0);
  const int64_t zbase = g * NZT;
  // row r of this launch -> atom (clamped rows: unconditional loads from valid addresses)
  auto atom = [&](int64_t r) __attribute__((always_inline)) -> int64_t {
    const int64_t rc = r < a.N ? r : a.N - 1;
    return a.perm != nullptr ? (int64_t)a.perm[rc] : rc;
  };
  const bool col_ok = (zlr < NZT) && (zbase + zl < a.N);
  const int64_t z = atom(zbase + zl);
  const int tzj = (a.types != nullptr && col_ok) ? (int)a.types[z] : 0;
  // atom types present in this unit (typed operand sets have at most 16 types): the stages of absent types are skipped
  unsigned present = 0xffffu;
  if (a.types != nullptr) {
    present = 0;
#pragma unroll
    for (int t = 0; t < 16; ++t) present |= (__any(col_ok && tzj == t) ? 1u : 0u) << t;
  }
  const int nct = (ch.mul_out + 31) / 32;
  const int ct0 = ch.c0 / 32;
  const bool two_tiles = cw > 32;

  int xz[XV4], xo[XV4];
#pragma unroll
  for (int v = 0; v < XV4; ++v) {
    const int idx = lane + v * 64;
    xz[v] = idx / RUN4;
    xo[v] = (idx - xz[v] * RUN4) * 4;
  }
  auto slot_ok = [&](int v) __attribute__((always_inline)) { return (v + 1) * 64 <= NZT * RUN4 || xz[v] < NZT; };
  // the atoms whose rows this lane stages, looked up ONCE per unit (with an atom order the row index is a load: inside the
  // stage loop it would put a memory round trip in front of every slab request)
  int zrow_x[XV4], zrow_g[XG4];
#pragma unroll
  for (int v = 0; v < XV4; ++v) zrow_x[v] = (int)atom(zbase + min(xz[v], NZT - 1));
#pragma unroll
  for (int v = 0; v < XG4; ++v) zrow_g[v] = (int)atom(zbase + min((lane + v * 64) / GQ, NZT - 1));

  auto fill = [&](NFStage& st) __attribute__((always_inline)) {
    const NFInstr& in = a.instr[__builtin_amdgcn_readfirstlane(st.q)];  // (scalar index: the tables stay in the kernel-argument segment)
    st.mul_in = in.mul_in;
    st.x_off = in.x_off;
    st.gate_off = in.gate_off;
    st.act = in.act;
    st.cst = in.cst;
    st.set = in.set;
    st.frag_off = in.frag_off;
    st.exp_off = in.exp_off;
    const NFSet& s = a.sets[__builtin_amdgcn_readfirstlane(in.set)];
    st.n_types = s.n_types;
    st.din = s.din;
    st.xp = s.x;
  };
  // (st.k0 == 0) move on to the next (instruction, type) whose type some atom of this unit has
  auto settle = [&](NFStage st) __attribute__((always_inline)) {
    while (st.valid && st.n_types > 1) {
      while (st.t < st.n_types && ((present >> st.t) & 1u) == 0u) ++st.t;
      if (st.t < st.n_types) break;
      st.t = 0;
      if (++st.q < ch.instr_end) fill(st);
      else st.valid = false;
    }
    return st;
  };
  auto first_stage = [&]() __attribute__((always_inline)) {
    NFStage st{};
    st.q = ch.instr_begin;
    st.valid = ch.instr_begin < ch.instr_end;
    if (st.valid) {
      fill(st);
      st = settle(st);
    }
    return st;
  };
  auto next_stage = [&](NFStage st) __attribute__((always_inline)) {
    if (!st.valid) return st;
    st.k0 += kNLK3;
    if (st.k0 >= st.mul_in) {
      st.k0 = 0;
      if (++st.t >= st.n_types) {
        st.t = 0;
        if (++st.q < ch.instr_end) fill(st);
        else st.valid = false;
      }
      st = settle(st);
    }
    return st;
  };

  // x slab of a stage: unconditional loads from clamped addresses (see node_linear_wave_bf16_unit)
  auto load_x = [&](float4 (&xr)[XV4], const NFStage& st) __attribute__((always_inline)) {
    const int kk = min(kNLK3, st.mul_in - st.k0) * D;
    const float* __restrict__ xb0 = st.xp + st.x_off + st.k0 * D;
    if (((st.din | st.x_off | kk) & 3) == 0) {
#pragma unroll
      for (int v = 0; v < XV4; ++v) {
        const int64_t zg = zrow_x[v];
        const int eo = min(xo[v], kk - 4);
        xr[v] = *reinterpret_cast<const float4*>(xb0 + zg * st.din + eo);
      }
    } else {
#pragma unroll
      for (int v = 0; v < XV4; ++v) {
        const int64_t zg = zrow_x[v];
        const float* __restrict__ p = xb0 + zg * st.din;
        xr[v].x = p[min(xo[v] + 0, kk - 1)];
        xr[v].y = p[min(xo[v] + 1, kk - 1)];
        xr[v].z = p[min(xo[v] + 2, kk - 1)];
        xr[v].w = p[min(xo[v] + 3, kk - 1)];
      }
    }
  };
  auto store_x = [&](const float4 (&xr)[XV4], const NFStage& st) __attribute__((always_inline)) {
    const int xkk = min(kNLK3, st.mul_in - st.k0) * D;
    const bool actv = st.gate_off == -1;  // (wave-uniform) activation of a scalar block while it is staged
    // (wave-uniform) a complete slab of a complete atom group needs no masking: everything a lane loaded is its own
    const bool whole = xkk == kNLK3 * D && zbase + NZT <= a.N;
#pragma unroll
    for (int v = 0; v < XV4; ++v) {
      if (slot_ok(v)) {
        float4 r = xr[v];
        if (actv) {
          r.x = nf_act(st.act, r.x, st.cst);
          r.y = nf_act(st.act, r.y, st.cst);
          r.z = nf_act(st.act, r.z, st.cst);
          r.w = nf_act(st.act, r.w, st.cst);
        }
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
  // gate scalars of a stage's channels: [NZT atoms][32 channels], activated once when they go to LDS.  (The host only asks
  // for gated blocks with d >= 3, multiplicities and offsets that are multiples of 4.)
  auto load_g = [&](float4 (&gr)[XG4], const NFStage& st) __attribute__((always_inline)) {
    if constexpr (D > 1) {
      if (st.gate_off >= 0) {
        const int kc = min(kNLK3, st.mul_in - st.k0);
        const float* __restrict__ gb0 = st.xp + st.gate_off + st.k0;
#pragma unroll
        for (int v = 0; v < XG4; ++v) {
          const int idx = lane + v * 64;
          const int gz = idx / GQ, go = (idx - gz * GQ) * 4;
          const int64_t zg = zrow_g[v];
          gr[v] = *reinterpret_cast<const float4*>(gb0 + zg * st.din + min(go, kc - 4));
        }
      }
    }
  };
  auto store_g = [&](const float4 (&gr)[XG4], const NFStage& st) __attribute__((always_inline)) {
    if constexpr (D > 1) {
      if (st.gate_off >= 0) {
        const int kc = min(kNLK3, st.mul_in - st.k0);
#pragma unroll
        for (int v = 0; v < XG4; ++v) {
          const int idx = lane + v * 64;
          const int gz = idx / GQ, go = (idx - gz * GQ) * 4;
          if (gz < NZT) {
            const bool zok = zbase + gz < a.N;
            float4 r;
            r.x = (zok && go + 0 < kc) ? nf_act(st.act, gr[v].x, st.cst) : 0.f;
            r.y = (zok && go + 1 < kc) ? nf_act(st.act, gr[v].y, st.cst) : 0.f;
            r.z = (zok && go + 2 < kc) ? nf_act(st.act, gr[v].z, st.cst) : 0.f;
            r.w = (zok && go + 3 < kc) ? nf_act(st.act, gr[v].w, st.cst) : 0.f;
            *reinterpret_cast<float4*>(gs + gz * GS + go) = r;
          }
        }
      }
    }
  };

  nl_u32x4 Af[2][2];  // [tile][plane]
  int we_blk = 0;
  auto load_a = [&](const NFStage& st, int k16) __attribute__((always_inline)) {
    const NFSet& s = a.sets[__builtin_amdgcn_readfirstlane(st.set)];
    const nl_u32x4* __restrict__ p = s.wf + (int64_t)st.t * s.frag_stride + st.frag_off + lane +
                                     ((int64_t)(k16 * nct + ct0) * 2) * 64;
    Af[0][0] = p[0]; Af[0][1] = p[64];
    if (two_tiles) { Af[1][0] = p[128]; Af[1][1] = p[192]; }
    we_blk = s.wexp[st.t * s.exp_stride + st.exp_off + k16];
  };
  f32x16n acc0 = {0}, acc1 = {0};
  constexpr int kUnset = 1 << 20;
  int Scol = kUnset;
  auto block = [&](int s, bool on, bool gated, bool masked) __attribute__((always_inline)) {
    const float* __restrict__ xb = xs + zl * S + m;
    float bq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bq[e] = xb[(16 * s + 8 * half + e) * D];
    if constexpr (D > 1) {
      if (gated) {  // (wave-uniform) act(gate[u]) of this lane's atom, channels 16 s + 8 half + e
        const float4 g0 = *reinterpret_cast<const float4*>(gs + zl * GS + 16 * s + 8 * half);
        const float4 g1 = *reinterpret_cast<const float4*>(gs + zl * GS + 16 * s + 8 * half + 4);
        bq[0] *= g0.x; bq[1] *= g0.y; bq[2] *= g0.z; bq[3] *= g0.w;
        bq[4] *= g1.x; bq[5] *= g1.y; bq[6] *= g1.z; bq[7] *= g1.w;
      }
    }
    // Columns are independent in the product, and a column that is never stored (lanes beyond the group's atoms: they read a
    // clamped atom's rows) may hold anything; rows beyond the slab and atoms beyond N were zero-filled when the slab was
    // staged.  Only a TYPED stage has to silence columns: atoms whose type is not the staged weight set's.
    if (masked) {
#pragma unroll
      for (int e = 0; e < 8; ++e) bq[e] = on ? bq[e] : 0.f;
    }
    const int we = we_blk;
    float mx = 0.f;
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
    if (__any(shift > 0)) {
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
    if (two_tiles) {
      acc0 = nl_mfma_f16(Af[0][1], Bh, acc0);
      acc1 = nl_mfma_f16(Af[1][1], Bh, acc1);
      acc0 = nl_mfma_f16(Af[0][0], Bl, acc0);
      acc1 = nl_mfma_f16(Af[1][0], Bl, acc1);
      acc0 = nl_mfma_f16(Af[0][0], Bh, acc0);
      acc1 = nl_mfma_f16(Af[1][0], Bh, acc1);
    } else {
      acc0 = nl_mfma_f16(Af[0][1], Bh, acc0);
      acc0 = nl_mfma_f16(Af[0][0], Bl, acc0);
      acc0 = nl_mfma_f16(Af[0][0], Bh, acc0);
    }
  };

  float4 xr0[XV4];
  float4 gr0[XG4];
  NFStage cur = first_stage();
  NFStage ld = cur;
  if constexpr (PIPE) {
    nl_u32x4 Ap[2][2][2][2];  // [buffer][K block][tile][plane]
    int wep[2][2];
    auto load_a2 = [&](int b, const NFStage& st) __attribute__((always_inline)) {
      const NFSet& s = a.sets[__builtin_amdgcn_readfirstlane(st.set)];
      const int k16a = st.k0 >> 4;
      const bool second = st.k0 + 16 < st.mul_in;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const int k16 = (kb == 1 && second) ? k16a + 1 : k16a;
        const nl_u32x4* __restrict__ p = s.wf + (int64_t)st.t * s.frag_stride + st.frag_off + lane +
                                         ((int64_t)(k16 * nct + ct0) * 2) * 64;
        Ap[b][kb][0][0] = p[0]; Ap[b][kb][0][1] = p[64];
        if (two_tiles) { Ap[b][kb][1][0] = p[128]; Ap[b][kb][1][1] = p[192]; }
        wep[b][kb] = s.wexp[__builtin_amdgcn_readfirstlane(st.t * s.exp_stride + st.exp_off + k16)];
      }
    };
    auto blocks = [&](int b, const NFStage& st) __attribute__((always_inline)) {
      const bool bsel = col_ok && (st.n_types == 1 || tzj == st.t);
      const bool gated = st.gate_off >= 0;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int tl2 = 0; tl2 < 2; ++tl2) {
          Af[tl2][0] = Ap[b][kb][tl2][0];
          Af[tl2][1] = Ap[b][kb][tl2][1];
        }
        we_blk = wep[b][kb];
        block(kb, bsel, gated, st.n_types > 1);
      }
    };
    if (ld.valid) { load_x(xr0, ld); load_g(gr0, ld); load_a2(0, ld); ld = next_stage(ld); }
    asm volatile("" ::"v"(tzj));  // (consume the atom-type load once, outside the loop: see node_linear_wave_bf16_unit)
    while (cur.valid) {
      store_x(xr0, cur);
      store_g(gr0, cur);
      __builtin_amdgcn_sched_barrier(0);
      if (ld.valid) { load_x(xr0, ld); load_g(gr0, ld); load_a2(1, ld); }
      __builtin_amdgcn_sched_barrier(0);
      blocks(0, cur);
      cur = ld;
      ld = next_stage(ld);
      if (!cur.valid) break;
      store_x(xr0, cur);
      store_g(gr0, cur);
      __builtin_amdgcn_sched_barrier(0);
      if (ld.valid) { load_x(xr0, ld); load_g(gr0, ld); load_a2(0, ld); }
      __builtin_amdgcn_sched_barrier(0);
      blocks(1, cur);
      cur = ld;
      ld = next_stage(ld);
    }
  } else {
  if (ld.valid) { load_x(xr0, ld); load_g(gr0, ld); ld = next_stage(ld); }
  while (cur.valid) {
    store_x(xr0, cur);
    store_g(gr0, cur);
    const int k16a = cur.k0 >> 4;
    load_a(cur, k16a);
    __builtin_amdgcn_sched_barrier(0);
    if (ld.valid) { load_x(xr0, ld); load_g(gr0, ld); ld = next_stage(ld); }
    __builtin_amdgcn_sched_barrier(0);
    const bool bsel = col_ok && (cur.n_types == 1 || tzj == cur.t);
    const bool gated = cur.gate_off >= 0;
#pragma unroll
    for (int s16 = 0; s16 < kNLK3 / 16; ++s16) {
      const bool exists = cur.k0 + 16 * s16 < cur.mul_in;
      (void)exists;
      block(s16, bsel, gated, cur.n_types > 1);
      if (s16 + 1 < kNLK3 / 16) {
        __builtin_amdgcn_sched_barrier(0);
        const bool nexists = cur.k0 + 16 * (s16 + 1) < cur.mul_in;
        load_a(cur, nexists ? k16a + s16 + 1 : k16a);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    cur = next_stage(cur);
  }
  }

  // ---- epilogue: accumulators -> the wavefront's slab -> contiguous runs per atom, through the chunk's epilogue
  constexpr int SE = kNLW * D + P;
  constexpr bool kVecE = (SE % 4) == 0;
  constexpr int RUN4E = kNLW * D / 4;
  constexpr int XV4E = (NZT * RUN4E + 63) / 64;
  static_assert(NZT * SE <= kNLXS, "result slab too small");
  {
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
  const NFDst& dst = a.dsts[ch.dst];
  const float scale = dst.scale;
  const int run = cw * D;
  if (ch.ep == 0) {
    const bool oal = ((dst.dout | ch.o_off | (ch.c0 * D)) & 3) == 0;
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
      const int64_t o = zg * dst.dout + ch.o_off + (int64_t)ch.c0 * D + eo;
      if (fast) {
        float4 r = *reinterpret_cast<const float4*>(sp);
        r.x *= scale; r.y *= scale; r.z *= scale; r.w *= scale;
        if (dst.addend != nullptr) {
          const float4 ad = *reinterpret_cast<const float4*>(dst.addend + o);
          r.x += ad.x; r.y += ad.y; r.z += ad.z; r.w += ad.w;
        }
        *reinterpret_cast<float4*>(dst.out + o) = r;
      } else if (zr < a.N && eo < run) {
        float4 r;
        if constexpr (kVecE) {
          r = *reinterpret_cast<const float4*>(sp);
        } else {
          r = make_float4(sp[0], sp[1], sp[2], sp[3]);
        }
        r.x *= scale; r.y *= scale; r.z *= scale; r.w *= scale;
        if (oal && eo + 3 < run) {
          if (dst.addend != nullptr) {
            const float4 ad = *reinterpret_cast<const float4*>(dst.addend + o);
            r.x += ad.x; r.y += ad.y; r.z += ad.z; r.w += ad.w;
          }
          *reinterpret_cast<float4*>(dst.out + o) = r;
        } else {
          const float rv[4] = {r.x, r.y, r.z, r.w};
          for (int e = 0; e < 4; ++e)
            if (eo + e < run) dst.out[o + e] = rv[e] + (dst.addend != nullptr ? dst.addend[o + e] : 0.f);
        }
      }
    }
  } else if (ch.ep == 1) {
    // scalar block of a gate's backward: grad_h[z, s] = r[z, s] * act'(h[z, s]); the host guarantees 16-byte alignment
    // (multiplicities, offsets and row strides are multiples of 4)
#pragma unroll
    for (int v = 0; v < XV4E; ++v) {
      const int idx = lane + v * 64;
      const int ez = idx / RUN4E;
      const int eo = (idx - ez * RUN4E) * 4;
      if (ez >= NZT || zbase + ez >= a.N || eo >= run) continue;
      const int64_t zg = atom(zbase + ez);
      const float* __restrict__ sp = xs + ez * SE + eo;
      float4 r;
      if constexpr (kVecE) {
        r = *reinterpret_cast<const float4*>(sp);
      } else {
        r = make_float4(sp[0], sp[1], sp[2], sp[3]);
      }
      const int64_t col = ch.ev_off + (int64_t)ch.c0 * D + eo;
      const float4 hv = *reinterpret_cast<const float4*>(a.h + zg * a.hdim + col);
      r.x *= scale * nf_act_grad(ch.act, hv.x, ch.cst);
      r.y *= scale * nf_act_grad(ch.act, hv.y, ch.cst);
      r.z *= scale * nf_act_grad(ch.act, hv.z, ch.cst);
      r.w *= scale * nf_act_grad(ch.act, hv.w, ch.cst);
      *reinterpret_cast<float4*>(dst.out + zg * dst.dout + col) = r;
    }
  } else {
    // gated block of a gate's backward, four channels (u .. u+3) of one atom per item:
    //   grad_h[z, v_u, m] = r[z, u, m] * act(h[z, q_u]);    grad_h[z, q_u] = act'(h[z, q_u]) * sum_m r[z, u, m] h[z, v_u, m]
    if constexpr (D > 1) {
      constexpr int NG = kNLW / 4;                     // channel groups per chunk
      constexpr int ROUNDS = (NZT * NG + 63) / 64;
#pragma unroll
      for (int it = 0; it < ROUNDS; ++it) {
        const int idx = lane + it * 64;
        const int ez = idx / NG;
        const int u0 = (idx - ez * NG) * 4;
        if (ez >= NZT || zbase + ez >= a.N || u0 >= cw) continue;
        const int64_t zg = atom(zbase + ez);
        const float* __restrict__ sp = xs + ez * SE + u0 * D;
        const float* __restrict__ hrow = a.h + zg * a.hdim;
        float* __restrict__ orow = dst.out + zg * dst.dout;
        const int64_t vcol = ch.ev_off + (int64_t)(ch.c0 + u0) * D;
        const int64_t gcol = ch.eg_off + ch.c0 + u0;
        float rr[4 * D], hh[4 * D];
#pragma unroll
        for (int q4 = 0; q4 < D; ++q4) {
          const float4 rv = *reinterpret_cast<const float4*>(sp + 4 * q4);
          const float4 hv = *reinterpret_cast<const float4*>(hrow + vcol + 4 * q4);
          rr[4 * q4 + 0] = rv.x * scale; rr[4 * q4 + 1] = rv.y * scale; rr[4 * q4 + 2] = rv.z * scale; rr[4 * q4 + 3] = rv.w * scale;
          hh[4 * q4 + 0] = hv.x; hh[4 * q4 + 1] = hv.y; hh[4 * q4 + 2] = hv.z; hh[4 * q4 + 3] = hv.w;
        }
        const float4 gt = *reinterpret_cast<const float4*>(hrow + gcol);
        const float gq[4] = {gt.x, gt.y, gt.z, gt.w};
        float go[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float ag = nf_act(ch.act, gq[c], ch.cst);
          float s = 0.f;
#pragma unroll
          for (int mm = 0; mm < D; ++mm) {
            s += rr[c * D + mm] * hh[c * D + mm];
            rr[c * D + mm] *= ag;
          }
          go[c] = s * nf_act_grad(ch.act, gq[c], ch.cst);
        }
#pragma unroll
        for (int q4 = 0; q4 < D; ++q4)
          *reinterpret_cast<float4*>(orow + vcol + 4 * q4) = make_float4(rr[4 * q4], rr[4 * q4 + 1], rr[4 * q4 + 2], rr[4 * q4 + 3]);
        *reinterpret_cast<float4*>(orow + gcol) = make_float4(go[0], go[1], go[2], go[3]);
      }
    }
  }
}

template <bool PIPE>
__global__ __launch_bounds__(64 * kNLWavesPerWG, PIPE ? 2 : 3) void node_fused_kernel(const NodeFusedArgs a_kernarg) {
  __shared__ __align__(16) float xs_all[kNLWavesPerWG * (kNLXS + kNFGS)];
  // The tables are read where the dispatch packet put them (the kernel-argument segment; this kernel's only argument sits at
  // its start): with the by-value parameter the compiler kept a 3 KiB private copy of the struct per lane -- scratch stores
  // at the top of every wavefront and scratch loads for every table access -- once the stage loop was unrolled.
  const NodeFusedArgs& a = *(const NodeFusedArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int unit = (int)blockIdx.x * kNLWavesPerWG + wv;
  if (unit >= a.grp_begin[a.n_groups]) return;
  float* xs = xs_all + wv * (kNLXS + kNFGS);
  float* gs = xs + kNLXS;
  int gi = 0;
  while (gi + 1 < a.n_groups && unit >= a.grp_begin[gi + 1]) ++gi;
  const int local = unit - a.grp_begin[gi];
  const int n = a.grp_n[gi];
  const NFChunk ch = a.chunks[a.grp_chunk0[gi] + local % n];
  const int64_t g = (int64_t)(local / n);
  switch (ch.d) {
    case 1: node_fused_unit<1, PIPE>(a, ch, g, xs, gs); break;
    case 3: node_fused_unit<3, PIPE>(a, ch, g, xs, gs); break;
    case 5: node_fused_unit<5, PIPE>(a, ch, g, xs, gs); break;
    case 7: node_fused_unit<7, PIPE>(a, ch, g, xs, gs); break;
    case 9: node_fused_unit<9, PIPE>(a, ch, g, xs, gs); break;
    default: break;
  }
}

}  // namespace nqa

// ---- host side: merge the parts of a launch into the kernel's tables --------------------------------------------------
namespace {

struct NFPlanned {
  std::vector<nqa::NFChunk> chunks;
  std::vector<nqa::NFInstr> instr;
  std::vector<int> stages;  // per chunk
};

const nqa_gate_block* nf_find_block(const nqa_gate_block* blocks, int32_t n, int32_t off, int32_t d, int32_t mul) {
  for (int32_t b = 0; b < n; ++b) {
    const nqa_gate_block& g = blocks[b];
    if (g.d == d && off >= g.out_off && (off - g.out_off) % d == 0 && off + mul * d <= g.out_off + g.mul * g.d) return &g;
  }
  return nullptr;
}

// returns "" or an error text
std::string nf_plan(const nqa_node_part* parts, int32_t n_parts, const nqa_gate_block* out_gate, int32_t n_out_gate,
                    NFPlanned& P) {
  using namespace nqa;
  if (n_parts < 1 || n_parts > 2) return "1 or 2 parts per launch";
  int32_t frag_off[2][kMaxNodeInstr + 1], mul_out[kMaxNodeInstr], exp_off[2][kMaxNodeInstr + 1];
  for (int p = 0; p < n_parts; ++p) {
    const nqa_node_part& pt = parts[p];
    if (pt.n_instr < 0 || pt.n_instr > kMaxNodeInstr || pt.n_chunks < 1 || pt.n_types < 1 || pt.dim_in <= 0 ||
        !pt.chunk_table || (pt.n_instr > 0 && !pt.instr_table))
      return "invalid part";
    const NodeChunk* chunks = static_cast<const NodeChunk*>(pt.chunk_table);
    const NodeInstr* instr = static_cast<const NodeInstr*>(pt.instr_table);
    if (node_frag_layout(chunks, pt.n_chunks, instr, pt.n_instr, frag_off[p], mul_out, 2) < 0) return "inconsistent tables";
    (void)node_exp_layout(instr, pt.n_instr, exp_off[p]);
  }
  const bool merged = n_parts == 2 && parts[1].accumulate != 0;
  if (merged) {
    const nqa_node_part &A = parts[0], &B = parts[1];
    if (A.n_chunks != B.n_chunks) return "accumulating parts must have the same output chunks";
    if (A.scale != B.scale) return "accumulating parts must share one scale (fold it into the weights)";
    const NodeChunk* ca = static_cast<const NodeChunk*>(A.chunk_table);
    const NodeChunk* cb = static_cast<const NodeChunk*>(B.chunk_table);
    for (int c = 0; c < A.n_chunks; ++c)
      if (ca[c].o_off != cb[c].o_off || ca[c].d != cb[c].d || ca[c].mul_out != cb[c].mul_out || ca[c].c0 != cb[c].c0)
        return "accumulating parts must have the same output chunks";
  }
  // instruction ranges: one contiguous range per distinct combination of source ranges
  struct Key { int b0, e0, b1, e1, begin, end; };
  std::vector<Key> ranges;
  auto add_instr = [&](int p, int q, std::string& err) {
    const nqa_node_part& pt = parts[p];
    const NodeInstr& in = static_cast<const NodeInstr*>(pt.instr_table)[q];
    NFInstr r{};
    r.x_off = in.x_off;
    r.mul_in = in.mul_in;
    r.frag_off = frag_off[p][q];
    r.exp_off = exp_off[p][q];
    r.gate_off = -2;
    r.act = 0;
    r.cst = 1.f;
    r.set = p;
    return r;
  };
  auto range_of = [&](int b0, int e0, int b1, int e1, int d, std::string& err) -> std::pair<int, int> {
    for (const Key& k : ranges)
      if (k.b0 == b0 && k.e0 == e0 && k.b1 == b1 && k.e1 == e1) return {k.begin, k.end};
    const int begin = (int)P.instr.size();
    for (int pass = 0; pass < 2; ++pass) {
      const int p = pass, b = pass == 0 ? b0 : b1, e = pass == 0 ? e0 : e1;
      for (int q = b; q < e; ++q) {
        NFInstr r = add_instr(p, q, err);
        const nqa_node_part& pt = parts[p];
        if (pt.in_gate != nullptr) {
          const nqa_gate_block* g = nf_find_block(static_cast<const nqa_gate_block*>(pt.in_gate), pt.n_in_gate, r.x_off, d, r.mul_in);
          if (g == nullptr) { err = "an instruction's input block is not a block of the input gate"; return {0, 0}; }
          const int rel = r.x_off - g->out_off;
          r.x_off = g->val_off + rel;
          r.act = g->act;
          r.cst = (float)g->cst;
          if (g->gate_off < 0) {
            r.gate_off = g->act == 0 ? -2 : -1;
          } else {
            r.gate_off = g->gate_off + rel / d;
            if (d == 1 || ((r.gate_off | r.mul_in | pt.dim_in) & 3) != 0) { err = "gated input blocks need d >= 3 and multiples of 4"; return {0, 0}; }
          }
        }
        P.instr.push_back(r);
      }
    }
    ranges.push_back({b0, e0, b1, e1, begin, (int)P.instr.size()});
    return {begin, (int)P.instr.size()};
  };
  std::string err;
  for (int p = 0; p < n_parts; ++p) {
    if (merged && p == 1) break;
    const nqa_node_part& pt = parts[p];
    const NodeChunk* chunks = static_cast<const NodeChunk*>(pt.chunk_table);
    for (int c = 0; c < pt.n_chunks; ++c) {
      const NodeChunk& cc = chunks[c];
      if (cc.d != 1 && cc.d != 3 && cc.d != 5 && cc.d != 7 && cc.d != 9) return "irrep dimension above 9 (l > 4)";
      if (cc.c0 % 64 != 0) return "chunks must start at multiples of 64 channels";
      NFChunk r{};
      r.o_off = cc.o_off;
      r.d = cc.d;
      r.mul_out = cc.mul_out;
      r.c0 = cc.c0;
      r.dst = merged ? 0 : p;
      std::pair<int, int> rg;
      if (merged) {
        const NodeChunk& cb = static_cast<const NodeChunk*>(parts[1].chunk_table)[c];
        rg = range_of(cc.instr_begin, cc.instr_end, cb.instr_begin, cb.instr_end, cc.d, err);
      } else if (p == 0) {
        rg = range_of(cc.instr_begin, cc.instr_end, 0, 0, cc.d, err);
      } else {
        rg = range_of(0, 0, cc.instr_begin, cc.instr_end, cc.d, err);
      }
      if (!err.empty()) return err;
      r.instr_begin = rg.first;
      r.instr_end = rg.second;
      r.ep = 0;
      r.act = 0;
      r.cst = 1.f;
      if (out_gate != nullptr) {
        const nqa_gate_block* g = nf_find_block(out_gate, n_out_gate, cc.o_off, cc.d, cc.mul_out);
        if (g == nullptr) return "an output block is not a block of the output gate";
        const int rel = cc.o_off - g->out_off;
        r.ev_off = g->val_off + rel;
        r.act = g->act;
        r.cst = (float)g->cst;
        if (g->gate_off < 0) {
          r.ep = 1;
          r.eg_off = -1;
          if (cc.d != 1 && g->act != 0) return "activated blocks of the output gate must be scalars";
          if (((r.ev_off | cc.mul_out | pt.dim_out) & 3) != 0) return "output gate: multiples of 4 required";
        } else {
          r.ep = 2;
          r.eg_off = g->gate_off + rel / cc.d;
          if (cc.d == 1 || ((r.ev_off | r.eg_off | cc.mul_out | pt.dim_out) & 3) != 0) return "output gate: gated blocks need d >= 3 and multiples of 4";
        }
      }
      int st = 0;
      for (int q = r.instr_begin; q < r.instr_end; ++q)
        st += parts[P.instr[q].set].n_types * ((P.instr[q].mul_in + nqa::kNLK3 - 1) / nqa::kNLK3);
      P.chunks.push_back(r);
      P.stages.push_back(st);
    }
  }
  if ((int)P.instr.size() > kNFMaxInstr) return "more than 48 instructions in one fused launch";
  return "";
}

}  // namespace

extern "C" {

int nqa_node_fused(const nqa_node_part* parts, int32_t n_parts, const int64_t* atom_types, const int32_t* atom_order,
                   int64_t num_nodes, const nqa_gate_block* out_gate, int32_t n_out_gate, const void* gate_h,
                   int32_t gate_dim, nqa_stream stream) {
  using namespace nqa;
  if (!node_f16()) {
    set_error("nqa_node_fused: needs the two-plane fp16 weight packing (NQA_NODE_F16=0 is set)");
    return NQA_ERR_UNSUPPORTED;
  }
  if (parts == nullptr || num_nodes < 0 || (out_gate != nullptr && (gate_h == nullptr || gate_dim <= 0 || n_out_gate < 1))) {
    set_error("nqa_node_fused: invalid argument");
    return NQA_ERR_INVALID;
  }
  NFPlanned P;
  const std::string err = nf_plan(parts, n_parts, out_gate, n_out_gate, P);
  if (!err.empty()) {
    set_error("nqa_node_fused: " + err);
    return NQA_ERR_INVALID;
  }
  if (num_nodes == 0) return NQA_OK;
  NodeFusedArgs a{};
  for (int p = 0; p < n_parts; ++p) {
    const nqa_node_part& pt = parts[p];
    if (!pt.x || !pt.packed || (pt.n_types > 1 && atom_types == nullptr)) {
      set_error("nqa_node_fused: NULL operand");
      return NQA_ERR_INVALID;
    }
    if (pt.n_types > 16) {
      set_error("nqa_node_fused: at most 16 atom types per typed operand set");
      return NQA_ERR_UNSUPPORTED;
    }
    int32_t fo[kMaxNodeInstr + 1], mo[kMaxNodeInstr], eo[kMaxNodeInstr + 1];
    const int64_t per_type = node_frag_layout(static_cast<const NodeChunk*>(pt.chunk_table), pt.n_chunks,
                                              static_cast<const NodeInstr*>(pt.instr_table), pt.n_instr, fo, mo, 2);
    NFSet& s = a.sets[p];
    s.x = static_cast<const float*>(pt.x);
    s.wf = static_cast<const nl_u32x4*>(pt.packed);
    s.frag_stride = per_type;
    s.exp_stride = node_exp_layout(static_cast<const NodeInstr*>(pt.instr_table), pt.n_instr, eo);
    s.wexp = reinterpret_cast<const int32_t*>(static_cast<const char*>(pt.packed) + per_type * 16 * pt.n_types);
    s.din = pt.dim_in;
    s.n_types = pt.n_types;
    if (!(p == 1 && pt.accumulate != 0)) {
      if (!pt.out || pt.dim_out <= 0) {
        set_error("nqa_node_fused: NULL destination");
        return NQA_ERR_INVALID;
      }
      NFDst& d = a.dsts[p];
      d.out = static_cast<float*>(pt.out);
      d.addend = static_cast<const float*>(pt.addend);
      d.dout = pt.dim_out;
      d.scale = (float)pt.scale;
      if (out_gate != nullptr && pt.dim_out != gate_dim) {
        set_error("nqa_node_fused: with an output gate the destination rows are the gate's input rows");
        return NQA_ERR_INVALID;
      }
    }
  }
  if (n_parts == 1) a.sets[1] = a.sets[0], a.dsts[1] = a.dsts[0];
  a.h = static_cast<const float*>(gate_h);
  a.hdim = gate_dim;
  a.types = atom_types;
  a.perm = atom_order;
  a.N = num_nodes;
  std::memcpy(a.instr, P.instr.data(), sizeof(NFInstr) * P.instr.size());
  const int n_chunks = (int)P.chunks.size();
  std::vector<int> order(n_chunks);
  for (int c = 0; c < n_chunks; ++c) order[c] = c;
  std::stable_sort(order.begin(), order.end(), [&](int l, int r) { return P.stages[l] > P.stages[r]; });
  hipStream_t s = static_cast<hipStream_t>(stream);
  for (int c0 = 0; c0 < n_chunks; c0 += kNFMaxChunks) {
    const int nc = std::min(n_chunks - c0, kNFMaxChunks);
    int64_t nblk = 0;
    int ng = 0;
    for (int c = 0; c < nc; ++c) {
      a.chunks[c] = P.chunks[order[c0 + c]];
      const NFChunk& cc = a.chunks[c];
      const bool same = c > 0 && cc.instr_begin == a.chunks[c - 1].instr_begin && cc.instr_end == a.chunks[c - 1].instr_end &&
                        cc.d == a.chunks[c - 1].d && cc.o_off == a.chunks[c - 1].o_off && cc.dst == a.chunks[c - 1].dst;
      const int per_unit = 32 / cc.d;
      if (!same) {
        a.grp_begin[ng] = (int32_t)nblk;
        a.grp_chunk0[ng] = c;
        a.grp_n[ng] = 0;
        ++ng;
      }
      ++a.grp_n[ng - 1];
      nblk += (num_nodes + per_unit - 1) / per_unit;
    }
    a.grp_begin[ng] = (int32_t)nblk;
    a.n_groups = ng;
    a.n_chunks = nc;
    if (nblk > 2147483647LL) {
      set_error("nqa_node_fused: too many work units for one launch");
      return NQA_ERR_UNSUPPORTED;
    }
    // NQA_NODE_PIPE=1: the one-wait-per-stage loop at two wavefronts per SIMD.  Measured (cfg-3 and cu20k, same box): no
    // faster -- these kernels are bound by how many vector instructions a stage issues with three wavefronts sharing a
    // SIMD, not by the waits of one wavefront, and the second fragment buffer costs the third wavefront.
    const char* pe = std::getenv("NQA_NODE_PIPE");
    if (pe != nullptr && pe[0] == '1')
      hipLaunchKernelGGL(node_fused_kernel<true>, dim3((unsigned)((nblk + kNLWavesPerWG - 1) / kNLWavesPerWG)),
                         dim3(64 * kNLWavesPerWG), 0, s, a);
    else
      hipLaunchKernelGGL(node_fused_kernel<false>, dim3((unsigned)((nblk + kNLWavesPerWG - 1) / kNLWavesPerWG)),
                         dim3(64 * kNLWavesPerWG), 0, s, a);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error(std::string("nqa_node_fused: ") + hipGetErrorString(e));
    return NQA_ERR_LAUNCH;
  }
  return NQA_OK;
}

// The merged tables of a launch, for host-side tests (no GPU): chunk records of 12 int32 (cst as float bits), instruction
// records of 8 int32.  Returns (n_chunks << 16) | n_instr, or -1 (nqa_last_error()).
int nqa_node_fused_plan(const nqa_node_part* parts, int32_t n_parts, const nqa_gate_block* out_gate, int32_t n_out_gate,
                        int32_t* chunks_out, int32_t chunks_cap, int32_t* instr_out, int32_t instr_cap) {
  NFPlanned P;
  const std::string err = nf_plan(parts, n_parts, out_gate, n_out_gate, P);
  if (!err.empty()) {
    set_error("nqa_node_fused_plan: " + err);
    return -1;
  }
  if (chunks_out != nullptr && (int32_t)(P.chunks.size() * 12) <= chunks_cap)
    std::memcpy(chunks_out, P.chunks.data(), P.chunks.size() * sizeof(nqa::NFChunk));
  if (instr_out != nullptr && (int32_t)(P.instr.size() * 8) <= instr_cap)
    std::memcpy(instr_out, P.instr.data(), P.instr.size() * sizeof(nqa::NFInstr));
  return (int)((P.chunks.size() << 16) | P.instr.size());
}

}  // extern "C"
