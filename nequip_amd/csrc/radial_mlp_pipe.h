// Internal (included by radial_mlp.hip): round-5 forms of the f16x3 radial-MLP kernels whose main loops are written for
// instruction ISSUE, not only for the matrix pipe.
//
// Why (profiles/r5_mlp_counters.txt, rocprofv3 SQ counters of the round-4 kernels alone): a wavefront of
// radial_mlp_fwd_split_bal_kernel<128, true> spends 35 % of its cycles issuing (6 vector + 2.5 scalar instructions per
// matrix instruction: 64-bit address arithmetic redone per unit, an integer modulo for the tile ring, exec-masked
// branches around every bounds check, IEEE division in SiLU), 43 % waiting for an issue slot / the matrix pipe and 22 %
// parked; the matrix pipe of a SIMD is busy 36 % of the time although its two wavefronts could fill it, because each
// wavefront's non-MFMA work sits in long blocks BETWEEN its MFMA groups (tile epilogue, staging), where the pipe idles
// unless the other wavefront happens to be in a matrix phase.  Here the hot loop is one branch-free region per work unit:
// every row base is wave-uniform (scalar registers) + a loop-invariant 32-bit lane offset, the weight-tile ring and the
// tile scales need no memory instruction or division, the epilogue of the previous tile is cut into pieces that ride in
// the shadows of the current tile's matrix instructions, only complete 32-column tiles are handled (other widths stay on
// the general kernel) and a ragged last block costs one wave-uniform test per unit.  Arithmetic, operand split, fragment layout (Wf, tile_scale of
// radial_mlp_split_w1_fwd_f16_kernel) and work decomposition are those of the kernel it replaces.
#pragma once

#include <type_traits>

namespace nqa {

// silu(x) = x / (1 + e^-x) on v_exp_f32 / v_rcp_f32 (1 ulp each; as the fused node stage, csrc/node_fused.h)
__device__ __forceinline__ float silu_fast_f(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}

// false for every valid launch, opaque to the compiler (ablation branches)
__device__ __forceinline__ bool ntiles_never(int W) { return W < 0; }

__device__ __forceinline__ float mlp_readlane_f(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

// Preconditions (checked by the host): E > 0, W % 32 == 0, W / 32 <= 128, nb <= kMaxNb.  The last block may be ragged:
// its missing rows compute on row E - 1 and are not stored (one wave-uniform test per unit).
template <int H, bool ABL = false, bool DIRECT = true>
__global__ __launch_bounds__(256, 2) void radial_mlp_fwd_pipe_kernel(const float* __restrict__ emb,
                                                                     const float* __restrict__ W0,
                                                                     const u32x4* __restrict__ Wf, float a0, int nb,
                                                                     int W, int64_t E, float* __restrict__ out,
                                                                     const float* __restrict__ tile_scale, int dbg_arg) {
  // ABL (NQA_MLP_DBG != 0; timing ablations, wrong results): the pieces named by the bits sit behind wave-uniform branches
  // that are never taken at run time, so that everything feeding them stays alive: 1 = no global stores, 2 = no matrix
  // instructions, 4 = no weight-tile staging after the first tile, 8 = no workgroup barrier, 16 = the hidden layer of the
  // first block for every block, 32 = the first k-step's LDS fragments for every k-step
  const int dbg = ABL ? dbg_arg : 0;
  const bool never = ABL && ntiles_never(W);
  constexpr int KS = H / 16;
  constexpr int TILE = KS * 2 * 64;  // uint4 per weight tile (two planes)
  constexpr int NV = TILE / 256;
  constexpr int kTS = 36;
  static_assert(TILE % 256 == 0, "tile must divide evenly over the workgroup");
  __shared__ u32x4 as[2][TILE];
  __shared__ __align__(16) float tbuf[4 * 32 * kTS];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const int l31 = lane & 31;
  const int ntiles = W >> 5;
  const int64_t nblk = (E + kMlpRows - 1) / kMlpRows;
  const int64_t U = nblk * ntiles;
  const int64_t u0 = U * (int64_t)blockIdx.x / gridDim.x;
  const int64_t u1 = U * ((int64_t)blockIdx.x + 1) / gridDim.x;
  if (u0 >= u1) return;  // (workgroup-uniform)
  const int n = (int)(u1 - u0);
  int64_t blk = u0 / ntiles;
  int t = (int)(u0 - blk * ntiles);

  // tile scales: lane l keeps tile l and tile l + 64, a unit reads its own with v_readlane (no memory instruction whose
  // wait would drain the LDS queue)
  const float tsA = tile_scale[lane < ntiles ? lane : ntiles - 1];
  const float tsB = tile_scale[lane + 64 < ntiles ? lane + 64 : ntiles - 1];
  auto tscale = [&](int tile) { return tile < 64 ? mlp_readlane_f(tsA, tile) : mlp_readlane_f(tsB, tile - 64); };

  auto load_ev = [&](int64_t b, float (&ev)[kMaxNb]) __attribute__((always_inline)) {
    const int64_t bc = b < nblk ? b : nblk - 1;
    const int64_t row = bc * kMlpRows + wv * 32 + l31;
    const float* __restrict__ er = emb + (row < E ? row : E - 1) * nb;
    if (nb == kMaxNb) {
      const float4 e0 = *reinterpret_cast<const float4*>(er);
      const float4 e1 = *reinterpret_cast<const float4*>(er + 4);
      ev[0] = e0.x; ev[1] = e0.y; ev[2] = e0.z; ev[3] = e0.w;
      ev[4] = e1.x; ev[5] = e1.y; ev[6] = e1.z; ev[7] = e1.w;
    } else {
#pragma unroll
      for (int c = 0; c < kMaxNb; ++c) ev[c] = er[c < nb ? c : nb - 1];
#pragma unroll
      for (int c = 0; c < kMaxNb; ++c) ev[c] = c < nb ? ev[c] : 0.f;
    }
  };

  u32x4 pre[NV];
  auto stage_load = [&](int tile) __attribute__((always_inline)) {
    const u32x4* __restrict__ src = Wf + (int64_t)tile * TILE;  // wave-uniform base
#pragma unroll
    for (int v = 0; v < NV; ++v) pre[v] = src[tid + v * 256];
  };
  auto stage_store = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int v = 0; v < NV; ++v) as[buf][tid + v * 256] = pre[v];
  };
  auto tile_after = [&](int tt, int k) {  // (tt + k) mod ntiles for k <= 2 without a division
    int x = tt + k;
    x = x >= ntiles ? x - ntiles : x;
    x = x >= ntiles ? x - ntiles : x;
    return ntiles == 1 ? 0 : x;
  };

  float ev[kMaxNb], evn[kMaxNb];
  load_ev(blk, ev);
  load_ev(blk + 1, evn);
  stage_load(t);
  stage_store(0);
  __syncthreads();

  // first-layer weights of this lane's A-fragment slots (x a0, zero beyond the basis): fetched once, kept for every block
  // (per block they cost an exposed L2 round trip: 9-14 us per launch, profiles/r5_mlp_ablations.txt)
  float w0r[H / 32][kMaxNb / 2];
#pragma unroll
  for (int kb = 0; kb < H / 32; ++kb)
#pragma unroll
    for (int s2 = 0; s2 < kMaxNb / 2; ++s2) {
      const int c = 2 * s2 + half;
      const float v = W0[(c < nb ? c : 0) * H + kb * 32 + l31];
      w0r[kb][s2] = c < nb ? v * a0 : 0.f;
    }
  // hidden rows of this wavefront's 32 edges as split B fragments (radial_mlp_fwd_split_bal_kernel::hidden, rows all valid)
  u32x4 bh[KS], bl[KS];
  float row_scale = 1.f;
  float rsr[DIRECT ? 16 : 1];  // DIRECT: the scales of the 16 rows this lane's accumulator registers belong to
  auto hidden = [&](const float (&e)[kMaxNb]) __attribute__((always_inline)) {
    float hv[(H / 32) * 16];
#pragma unroll
    for (int kb = 0; kb < H / 32; ++kb) {
      f32x16 hacc = {0};
#pragma unroll
      for (int s2 = 0; s2 < kMaxNb / 2; ++s2) {
        const float av = w0r[kb][s2];
        const float bv = half ? e[2 * s2 + 1] : e[2 * s2];
        hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, hacc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) hv[kb * 16 + r] = silu_fast_f(hacc[r]);
    }
    float m = 0.f;  // the row's H values sit in this lane and in lane ^ 32
#pragma unroll
    for (int i = 0; i < (H / 32) * 16; ++i) m = fmaxf(m, fabsf(hv[i]));
    m = fmaxf(m, __shfl_xor(m, 32));
    const float su = f16_scale_up(m);
    row_scale = 1.f / su;
    if constexpr (DIRECT) {
#pragma unroll
      for (int r = 0; r < 16; ++r) rsr[r] = __shfl(row_scale, (r & 3) + 8 * (r >> 2) + 4 * half, 64);
    }
#pragma unroll
    for (int kb = 0; kb < H / 32; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        uint32_t a, b;
        split_pair_f16(hv[kb * 16 + r] * su, hv[kb * 16 + r + 1] * su, a, b);
        const int s = 2 * kb + (r >> 3), tp = (r & 7) >> 1;
        bh[s][tp] = a; bl[s][tp] = b;
      }
    }
  };
  hidden(ev);

  // loop-invariant lane parts of the epilogue addresses
  const int c4 = lane & 7, rsub = lane >> 3;
  float* tbw = tbuf + wv * (32 * kTS) + l31 * kTS + 4 * half;  // + 8 g: this lane's four columns of register quad g
  const float* tbr = tbuf + wv * (32 * kTS) + rsub * kTS + 4 * c4;  // + 8 i kTS: row 8 i + rsub, columns 4 c4 ..
  unsigned soff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    soff[i] = DIRECT ? ((unsigned)(8 * i + 4 * half) * (unsigned)W + (unsigned)l31) * 4u  // rows 8 i + 4 half (+ r & 3), column l31
                     : ((unsigned)(8 * i + rsub) * (unsigned)W + 4u * (unsigned)c4) * 4u;

  int prev_tile = 0;
  int64_t prev_row0 = 0;
  float prev_rs = 1.f;
  float fr[DIRECT ? 16 : 1];  // DIRECT: row scale x tile scale of the tile whose accumulators wait for their epilogue

  // One work unit.  FIRST: no previous tile to write out.  The epilogue of the previous tile rides in the shadows of this
  // tile's matrix instructions, a few instructions per k-step:
  //  DIRECT (default): the product is formed as (hidden rows) x (weight tile) -- the two fragment operands of the matrix
  //   instruction swapped, same data -- so that a lane holds ONE column and 16 rows: register r goes out as two complete
  //   128-byte row segments per store instruction, no pass through LDS (the LDS pipe is what the round-4 kernel saturated:
  //   16-byte LDS writes cost 13 cycles per wavefront instruction, profiles/r5_mlp_ablations.txt);
  //     s = 1 .. 6 : (pa + pb) * fr and the 16 dword stores, three registers per k-step        (H = 64: steps 0 .. 2)
  //     s = 6      : the next unit's weight tile, registers -> LDS                               (H = 64: step 3)
  //     s = 7      : fr of THIS tile for the next unit's epilogue
  //  !DIRECT (round-4 layout, lane = row): through a wave-private LDS transpose
  //     s = 1, 2 : (pa + pb) * f and the four 16-byte LDS writes; s = 3: four row-major LDS reads; s = 4, 5: four stores
  constexpr bool kLong = KS >= 8;             // H = 128: eight k-steps to spread the pieces over; H = 64: four
  constexpr int sW = kLong ? 1 : 0, nW = kLong ? 2 : 1, sR = sW + nW, sS = sR + 1, nS = kLong ? 2 : 1;
  constexpr int sG = kLong ? 6 : KS - 1;
  constexpr int sD = kLong ? 1 : 0, nD = kLong ? 6 : 3;  // DIRECT: first step and number of steps of the epilogue
  static_assert(sS + nS <= KS && sG < KS && sD + nD <= KS, "epilogue pieces must fit the k-steps");
  // rows_left: rows of the tile's 32 that exist (32 everywhere but in a ragged last block; wave-uniform)
  auto emit_direct = [&](const f32x16& pa, const f32x16& pb, float* ob, int r0, int r1, int rows_left)
      __attribute__((always_inline)) {
    if (rows_left >= 32) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (r < r0 || r >= r1) continue;
        const float v = (pa[r] + pb[r]) * fr[r];
        float* __restrict__ dst = reinterpret_cast<float*>(reinterpret_cast<char*>(ob + (int64_t)(r & 3) * W) + soff[r >> 2]);
        if (!(dbg & 1) || never) __builtin_nontemporal_store(v, dst);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (r < r0 || r >= r1) continue;
        const float v = (pa[r] + pb[r]) * fr[r];
        float* __restrict__ dst = reinterpret_cast<float*>(reinterpret_cast<char*>(ob + (int64_t)(r & 3) * W) + soff[r >> 2]);
        if ((r & 3) + 8 * (r >> 2) + 4 * half < rows_left) *dst = v;
      }
    }
  };
  auto store_rows = [&](float* ob, const float4 (&rd)[4], int q0, int q1, int rows_left) __attribute__((always_inline)) {
    if (rows_left >= 32) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (q >= q0 && q < q1 && (!(dbg & 1) || never))
          mlp_store4(reinterpret_cast<float*>(reinterpret_cast<char*>(ob) + soff[q]), rd[q]);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (q >= q0 && q < q1 && 8 * q + rsub < rows_left)
          *reinterpret_cast<float4*>(reinterpret_cast<char*>(ob) + soff[q]) = rd[q];
    }
  };
  auto unit = [&](int i, auto first_tag, f32x16& accA, f32x16& accB, const f32x16& prevA, const f32x16& prevB)
      __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_tag)::value;
    if (t == 0 && i > 0) {  // entering the next block (rare: once per ntiles units)
      ++blk;
#pragma unroll
      for (int c = 0; c < kMaxNb; ++c) ev[c] = evn[c];
      if (!(dbg & 16) || never) hidden(ev);
      load_ev(blk + 1, evn);
    }
    const int buf = i & 1;
    // The weight tile of unit i + 1 is requested here and goes to the other LDS buffer at k-step 6 of THIS unit, behind
    // the epilogue's stores in program order: inside one region the compiler counts the memory queue exactly (the
    // younger stores stay in flight).  Consumed at the top of the next unit instead, the wait sat behind a loop head
    // where the counts of all predecessors merge, and every unit waited for its own stores.
    if (!(dbg & 4) || never) stage_load(tile_after(t, 1));  // unconditional: past the range a valid tile is fetched and never used
    const u32x4* __restrict__ a = as[buf] + lane;
    accA = (f32x16){0};
    accB = (f32x16){0};
    float f = 0.f;
    float* __restrict__ ob = nullptr;
    int rows_left = 32;
    if constexpr (!FIRST) {
      if constexpr (!DIRECT) f = prev_rs * tscale(prev_tile);
      ob = out + prev_row0 * W + prev_tile * 32;  // wave-uniform
      rows_left = (int)(E - prev_row0 < 32 ? E - prev_row0 : 32);  // (<= 0: nothing of this wavefront's rows exists)
    }
    float4 rd[4];
    u32x4 fa[3][2];  // weight fragments two k-steps ahead (LDS latency under load exceeds one k-step of matrix work)
    fa[0][0] = a[0];
    fa[0][1] = a[64];
    if (KS > 1) {
      fa[1][0] = a[128];
      fa[1][1] = a[192];
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (s + 2 < KS && (!(dbg & 32) || never)) {
        fa[(s + 2) % 3][0] = a[((s + 2) * 2) * 64];
        fa[(s + 2) % 3][1] = a[((s + 2) * 2 + 1) * 64];
      }
      const u32x4 &ah = fa[s % 3][0], &al = fa[s % 3][1];
      if (ABL && (dbg & 2) && !never) {
      } else if constexpr (DIRECT) {
        if (s & 1) {
          accB = mfma_f16(bl[s], ah, accB);
          accA = mfma_f16(bh[s], al, accA);
          accB = mfma_f16(bh[s], ah, accB);
        } else {
          accA = mfma_f16(bl[s], ah, accA);
          accB = mfma_f16(bh[s], al, accB);
          accA = mfma_f16(bh[s], ah, accA);
        }
      } else if (s & 1) {
        accB = mfma_f16(ah, bl[s], accB);
        accA = mfma_f16(al, bh[s], accA);
        accB = mfma_f16(ah, bh[s], accB);
      } else {
        accA = mfma_f16(ah, bl[s], accA);
        accB = mfma_f16(al, bh[s], accB);
        accA = mfma_f16(ah, bh[s], accA);
      }
      if constexpr (!FIRST && DIRECT) {
        if (s >= sD && s < sD + nD) emit_direct(prevA, prevB, ob, (s - sD) * 16 / nD, (s - sD + 1) * 16 / nD, rows_left);
      }
      if constexpr (!FIRST && !DIRECT) {
        if (s >= sW && s < sW + nW) {
#pragma unroll
          for (int g = (s - sW) * 4 / nW; g < (s - sW + 1) * 4 / nW; ++g)
            *reinterpret_cast<float4*>(tbw + 8 * g) =
                make_float4((prevA[4 * g] + prevB[4 * g]) * f, (prevA[4 * g + 1] + prevB[4 * g + 1]) * f,
                            (prevA[4 * g + 2] + prevB[4 * g + 2]) * f, (prevA[4 * g + 3] + prevB[4 * g + 3]) * f);
        }
        if (s == sR) {
#pragma unroll
          for (int q = 0; q < 4; ++q) rd[q] = *reinterpret_cast<const float4*>(tbr + 8 * q * kTS);
        }
        if (s >= sS && s < sS + nS) store_rows(ob, rd, (s - sS) * 4 / nS, (s - sS + 1) * 4 / nS, rows_left);
      }
      if (s == sG && (!(dbg & 4) || never)) stage_store(buf ^ 1);
      if constexpr (DIRECT) {
        if (s == KS - 1) {
          const float tsc = tscale(t);
#pragma unroll
          for (int r = 0; r < 16; ++r) fr[r] = rsr[r] * tsc;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!(dbg & 8)) lds_barrier();
    prev_tile = t;
    prev_row0 = blk * kMlpRows + wv * 32;
    prev_rs = row_scale;
    t = (t + 1 == ntiles) ? 0 : t + 1;
  };

  f32x16 a0A, a0B, a1A, a1B;
  using first_t = std::integral_constant<bool, true>;
  using next_t = std::integral_constant<bool, false>;
  unit(0, first_t{}, a0A, a0B, a0A, a0B);  // (no previous tile: the last two arguments are unused)
  int i = 1;
  for (; i + 1 < n; i += 2) {
    unit(i, next_t{}, a1A, a1B, a0A, a0B);
    unit(i + 1, next_t{}, a0A, a0B, a1A, a1B);
  }
  const bool odd_left = i < n;  // one more unit (its accumulators: set 1)
  if (odd_left) unit(i, next_t{}, a1A, a1B, a0A, a0B);
  // epilogue of the last unit (not overlapped with anything)
  {
    const f32x16& pa = odd_left ? a1A : a0A;
    const f32x16& pb = odd_left ? a1B : a0B;
    float* __restrict__ ob = out + prev_row0 * W + prev_tile * 32;
    const int rows_left = (int)(E - prev_row0 < 32 ? E - prev_row0 : 32);
    if constexpr (DIRECT) {
      emit_direct(pa, pb, ob, 0, 16, rows_left);
    } else {
      const float f = prev_rs * tscale(prev_tile);
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(tbw + 8 * g) =
            make_float4((pa[4 * g] + pb[4 * g]) * f, (pa[4 * g + 1] + pb[4 * g + 1]) * f,
                        (pa[4 * g + 2] + pb[4 * g + 2]) * f, (pa[4 * g + 3] + pb[4 * g + 3]) * f);
      float4 rd[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) rd[q] = *reinterpret_cast<const float4*>(tbr + 8 * q * kTS);
      store_rows(ob, rd, 0, 4, rows_left);
    }
  }
}


// silu'(x) = s (1 + x (1 - s)), s = 1 / (1 + e^-x), on v_exp_f32 / v_rcp_f32
__device__ __forceinline__ float silu_grad_fast_f(float x) {
  const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
  return sg * (1.0f + x * (1.0f - sg));
}

// ---- f16x3 inference backward over BALANCED work-unit ranges -------------------------------------------------------------
// g_emb = ((g_w W1s^T) silu'(P)) W0s^T as radial_mlp_bwd_split_kernel<H, 0, false, true> computes it (same fragments Wb /
// chunk_exp of radial_mlp_split_w1_bwd_f16_kernel, same running per-row exponent, same epilogue), but
//  * the launch is a fixed grid of two workgroups per CU and workgroup g owns the contiguous range [g U / G, (g + 1) U / G) of the
//    U = (128-row blocks) x (32-column K chunks) work units in block-major order.  One workgroup per block ran the 1565 blocks
//    of the cfg-3 pair list as 3.06 rounds over the 512 slots: the fourth, almost empty round cost a quarter of the kernel
//    (the forward was balanced the same way in round 3).  A block whose chunks are shared by two workgroups gets two partial
//    g_h -- the epilogue is linear in g_h, so each workgroup finishes its own part and ADDS its g_emb rows to a zeroed
//    buffer (hardware float atomics; with at most two addends per element the sum does not depend on their order: bitwise
//    reproducible).  G <= number of blocks, hence a range holds at least one block's worth of chunks and no block has
//    three owners;
//  * only complete 128-row blocks and 32-column chunks are handled (a ragged tail goes to the general kernel), every row base
//    is wave-uniform + a 32-bit lane offset, and the loads of a chunk are unconditional (past the range: a valid chunk,
//    never used) -- the chunk body has no exec-masked region besides the rare rescale.
//    The last block may be ragged: its missing rows read row E - 1 again and are never written.
// Preconditions (host): E > 0, W % 32 == 0, gridDim.x <= ceil(E / kMlpRows), g_emb zero-filled.
template <int H, int PF = 2, bool RAGGED = true, bool ABL = false>
__global__ __launch_bounds__(256, 2) void radial_mlp_bwd_pipe_kernel(const float* __restrict__ emb,
                                                                     const float* __restrict__ W0,
                                                                     const u32x4* __restrict__ Wb,
                                                                     const float* __restrict__ gw, float a0, int nb,
                                                                     int W, int64_t E, float* __restrict__ g_emb,
                                                                     const int* __restrict__ chunk_exp, int dbg_arg) {
  // ABL (NQA_MLP_DBG_BWD != 0; timing ablations, wrong results; never-taken wave-uniform branches keep the operands alive):
  // 2 = no matrix instructions, 4 = no weight-fragment staging, 32 = no LDS fragment reads
  const int dbg = ABL ? dbg_arg : 0;
  const bool never = ABL && ntiles_never(W);
  constexpr int NT = H / 32;
  constexpr int CH = 2 * 2 * NT * 64;  // uint4 per chunk of B fragments (two k-steps x two planes)
  constexpr int NV = CH / 256;
  constexpr int GS = H + 1;
  constexpr int kMainBytes = 2 * CH * 16;
  constexpr int kEpiBytes = kMlpRows * GS * 4;
  constexpr int kBufBytes = kMainBytes > kEpiBytes ? kMainBytes : kEpiBytes;
  __shared__ __align__(16) unsigned char smem_raw[kBufBytes];
  __shared__ float w0s[H * kMaxNb];        // [k][c]
  __shared__ float w0t[kMaxNb * H];        // [c][k]
  __shared__ float es[kMlpRows * kMaxNb];  // embedding tile [row][c]
  __shared__ int rowexp[kMlpRows];
  u32x4* __restrict__ bsm = reinterpret_cast<u32x4*>(smem_raw);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const int l31 = lane & 31;
  const int nchunks = W >> 5;
  const int64_t nblk = (E + kMlpRows - 1) / kMlpRows;
  const int64_t U = nblk * nchunks;
  const int64_t u0 = U * (int64_t)blockIdx.x / gridDim.x;
  const int64_t u1 = U * ((int64_t)blockIdx.x + 1) / gridDim.x;
  if (u0 >= u1) return;  // (workgroup-uniform)

  for (int i = tid; i < H * kMaxNb; i += 256) {
    const int k = i / kMaxNb, c = i - k * kMaxNb;
    const float v = c < nb ? W0[c * H + k] * a0 : 0.f;
    w0s[i] = v;
    w0t[c * H + k] = v;
  }
  constexpr int kUnset = 1 << 20;
  const unsigned goff_full = ((unsigned)(wv * 32 + l31) * (unsigned)W + 16u * (unsigned)half) * 4u;

  // rows (r, half) of the accumulator registers <- values held by the rows' own lanes (same wavefront: LDS in order)
  auto rows_from_lanes = [&](int value, int (&out)[16]) __attribute__((always_inline)) {
    int* __restrict__ rb = rowexp + wv * 32;
    if (half == 0) rb[l31] = value;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int r = 0; r < 16; ++r) out[r] = rb[(r & 3) + 8 * (r >> 2) + 4 * half];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };

  int64_t u = u0;
  while (u < u1) {
    const int64_t blk = u / nchunks;
    const int c0 = (int)(u - blk * nchunks);
    const int n = (int)((u1 - u) < (int64_t)(nchunks - c0) ? (u1 - u) : (int64_t)(nchunks - c0));  // chunks of this segment
    const int c1 = c0 + n;
    u += n;
    const int64_t blk0 = blk * kMlpRows;
    const float* __restrict__ gbase = gw + blk0 * W;  // wave-uniform
    const int rows_here = RAGGED ? (int)(E - blk0 < (int64_t)kMlpRows ? E - blk0 : (int64_t)kMlpRows) : kMlpRows;
    // lane part of this lane's g_w row inside the block (bytes): row 32 wv + l31 (clamped into a ragged block), the 16 floats
    // 16 half .. of a chunk
    const int rl = wv * 32 + l31;
    const unsigned goff = RAGGED ? ((unsigned)(rl < rows_here ? rl : rows_here - 1) * (unsigned)W + 16u * (unsigned)half) * 4u
                                 : goff_full;

    __syncthreads();  // the previous segment's epilogue is done with smem_raw / es (first segment: w0s / w0t are in place)
    for (int i = tid; i < kMlpRows * kMaxNb; i += 256) {
      const int r = i / kMaxNb, c = i - r * kMaxNb;
      const float v = emb[(blk0 + (r < rows_here ? r : rows_here - 1)) * nb + (c < nb ? c : nb - 1)];
      es[i] = (c < nb && r < rows_here) ? v : 0.f;
    }

    u32x4 pb[NV];
    auto load_a = [&](int ch, float4 (&pa)[4]) __attribute__((always_inline)) {
      const int cc = ch < c1 ? ch : c1 - 1;  // past the segment: a valid chunk, never used
      const char* __restrict__ src = reinterpret_cast<const char*>(gbase + 32 * cc) + goff;
#pragma unroll
      for (int v = 0; v < 4; ++v) pa[v] = *reinterpret_cast<const float4*>(src + 16 * v);
    };
    auto load_b = [&](int ch) __attribute__((always_inline)) {
      const int cc = ch < c1 ? ch : c1 - 1;
      const u32x4* __restrict__ src = Wb + (int64_t)cc * CH;
#pragma unroll
      for (int v = 0; v < NV; ++v) pb[v] = src[tid + v * 256];
    };
    auto store_b = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
      for (int v = 0; v < NV; ++v) bsm[buf * CH + tid + v * 256] = pb[v];
    };
    // g_w rows (HBM) kPF chunks ahead in rotating register sets: with two sets a wavefront had 8 KiB in flight, 64 KiB per CU --
    // at the loaded latency of a saturated HBM (several microseconds) that caps the chip below 4 TB/s (Little's law)
    constexpr int kPF = PF;
    static_assert(kPF == 2 || kPF == 4, "prefetch depth");
    float4 pa[kPF][4];
    auto& pa_all = pa;
    load_b(c0);
    load_a(c0, pa[0]);
    store_b(0);
#pragma unroll
    for (int k = 1; k < kPF; ++k) load_a(c0 + k, pa[k]);
    __syncthreads();

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x16){0};
    int S = kUnset;               // exponent of this lane's row (unset while the row is all zero)
    int we_next = chunk_exp[c0];  // weight exponent of the chunk about to be consumed

    // kslot: compile-time position of the body in the unrolled group of kPF (PF == 4: which register sets an odd body refills)
    auto body = [&](int ch, int j, float4 (&pa)[4], auto kslot_tag) __attribute__((always_inline)) {
      constexpr int kslot = decltype(kslot_tag)::value;
      const int buf = j & 1;
      u32x4 ah[2], al[2];
      const int we = we_next;
      we_next = chunk_exp[ch + 1 < c1 ? ch + 1 : c1 - 1];
      float m = 0.f;  // the row's largest magnitude in this chunk (its other 16 values sit in lane ^ 32)
#pragma unroll
      for (int v = 0; v < 4; ++v)
        m = fmaxf(m, fmaxf(fmaxf(fabsf(pa[v].x), fabsf(pa[v].y)), fmaxf(fabsf(pa[v].z), fabsf(pa[v].w))));
      m = fmaxf(m, __shfl_xor(m, 32));
      int shift = 0;
      if (m > 0.f && m < 3.0e38f) {
        int em;
        (void)frexpf(m, &em);          // m < 2^em
        const int cap = 15 - em + we;  // largest S that keeps m 2^(S - we) below 2^15
        if (cap < S) {
          int ns = cap - 3;
          ns = ns > we + 100 ? we + 100 : (ns < we - 100 ? we - 100 : ns);
          shift = S == kUnset ? 0 : S - ns;
          S = ns;
        }
      }
      if (j > 0 && __any(shift > 0)) {  // (rare after the first chunks: a new maximum 8x above every earlier one)
        int kr[16];
        rows_from_lanes(shift, kr);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t][r] = ldexpf(acc[t][r], -kr[r]);
      }
      int q = S == kUnset ? 0 : S - we;
      q = q > 120 ? 120 : (q < -120 ? -120 : q);
      const float qs = ldexpf(1.f, q);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        uint32_t a, b;
        split_pair_f16(pa[2 * s].x * qs, pa[2 * s].y * qs, a, b);
        ah[s][0] = a; al[s][0] = b;
        split_pair_f16(pa[2 * s].z * qs, pa[2 * s].w * qs, a, b);
        ah[s][1] = a; al[s][1] = b;
        split_pair_f16(pa[2 * s + 1].x * qs, pa[2 * s + 1].y * qs, a, b);
        ah[s][2] = a; al[s][2] = b;
        split_pair_f16(pa[2 * s + 1].z * qs, pa[2 * s + 1].w * qs, a, b);
        ah[s][3] = a; al[s][3] = b;
      }
      // weight fragments (L2) of the next chunk BEFORE the g_w request (HBM) of a later chunk: vmcnt retires in order,
      // so the wait for the fragments at the end of this body leaves the four younger HBM loads in flight
      if (!(dbg & 4) || never) load_b(ch + 1);
      if constexpr (kPF == 4) {
        // PF == 4: the rows are requested in bursts of TWO adjacent chunks (256 contiguous bytes per row instead of 128) by
        // the odd bodies, once both register sets involved are free: chunks ch + 3 and ch + 4
        if constexpr ((kslot & 1) != 0) {
          load_a(ch + 3, pa_all[(kslot + 3) & 3]);
          load_a(ch + 4, pa_all[kslot]);
        }
      } else {
        load_a(ch + kPF, pa);
      }
      const u32x4* __restrict__ bs = bsm + buf * CH + lane;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        u32x4 fb[2][NT];
        if (!(dbg & 32) || never || (s == 0 && j == 0)) {
#pragma unroll
          for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int t = 0; t < NT; ++t) fb[p][t] = bs[((s * 2 + p) * NT + t) * 64];
        } else {
#pragma unroll
          for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int t = 0; t < NT; ++t) fb[p][t] = (u32x4){ah[s][0] + (unsigned)t, al[s][1], ah[s][2] + (unsigned)p, al[s][3]};
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ABL && (dbg & 2) && !never) {
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t][0] += __builtin_bit_cast(float, fb[0][t][0] ^ fb[1][t][1] ^ al[s][t & 3] ^ ah[s][(t + 1) & 3]);
        } else {
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = mfma_f16(al[s], fb[0][t], acc[t]);
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = mfma_f16(ah[s], fb[1][t], acc[t]);
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = mfma_f16(ah[s], fb[0][t], acc[t]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!(dbg & 4) || never) store_b(buf ^ 1);
      lds_barrier();
    };
    for (int j = 0; j < n; j += kPF) {
      body(c0 + j, j, pa[0], std::integral_constant<int, 0>{});
      if (kPF > 1 && j + 1 < n) body(c0 + j + 1, j + 1, pa[1 % kPF], std::integral_constant<int, 1>{});
      if (kPF > 2 && j + 2 < n) body(c0 + j + 2, j + 2, pa[2 % kPF], std::integral_constant<int, 2>{});
      if (kPF > 3 && j + 3 < n) body(c0 + j + 3, j + 3, pa[3 % kPF], std::integral_constant<int, 3>{});
    }

    {  // accumulators back to the true scale: 2^-S of their row
      int sr[16];
      rows_from_lanes(S == kUnset ? 0 : S, sr);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = ldexpf(acc[t][r], -sr[r]);
    }
    // epilogue (exact fp32): pre-activations recomputed on MFMA in the accumulator layout, g_pre = g_h silu'(pre), one pass
    // through LDS for the NB-wide GEMV.  (every wavefront is past the last chunk's barrier: the weight buffers are free)
    float* __restrict__ gp = reinterpret_cast<float*>(smem_raw);
    {
      const float* __restrict__ erow = es + (wv * 32 + l31) * kMaxNb;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        f32x16 pacc = {0};
#pragma unroll
        for (int s2 = 0; s2 < kMaxNb / 2; ++s2) {
          const float av = erow[2 * s2 + half];
          const float bv = w0t[(2 * s2 + half) * H + t * 32 + l31];
          pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, pacc, 0, 0, 0);
        }
        const int col = t * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int lr = wv * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          gp[lr * GS + col] = acc[t][r] * silu_grad_fast_f(pacc[r]);
        }
      }
    }
    __syncthreads();
    {
      const int r = tid >> 1, kh = tid & 1;
      float sacc[kMaxNb];
#pragma unroll
      for (int c = 0; c < kMaxNb; ++c) sacc[c] = 0.f;
      const float* __restrict__ gr = gp + r * GS + kh * (H / 2);
      const float* __restrict__ wr = w0s + kh * (H / 2) * kMaxNb;
#pragma unroll 4
      for (int k = 0; k < H / 2; ++k) {
        const float gv = gr[k];
        const float4 w0 = *reinterpret_cast<const float4*>(wr + k * kMaxNb);
        const float4 w1 = *reinterpret_cast<const float4*>(wr + k * kMaxNb + 4);
        sacc[0] += gv * w0.x; sacc[1] += gv * w0.y; sacc[2] += gv * w0.z; sacc[3] += gv * w0.w;
        sacc[4] += gv * w1.x; sacc[5] += gv * w1.y; sacc[6] += gv * w1.z; sacc[7] += gv * w1.w;
      }
#pragma unroll
      for (int c = 0; c < kMaxNb; ++c) sacc[c] += __shfl_xor(sacc[c], 1, 64);
      if (kh == 0 && r < rows_here) {
        float* __restrict__ dst = g_emb + (blk0 + r) * nb;
        if (n == nchunks) {  // the whole block: plain stores (nobody else adds to these rows)
          for (int c = 0; c < nb; ++c) dst[c] = sacc[c];
        } else {
          for (int c = 0; c < nb; ++c) unsafeAtomicAdd(dst + c, sacc[c]);
        }
      }
    }
  }
}


// ---- f16x3 inference backward, g_w read in COALESCED 256-byte row pieces --------------------------------------------------
// What bounds radial_mlp_bwd_pipe_kernel (and the round-4 kernel) is how its A operand arrives: lane = row, i.e. every load
// instruction of a wavefront touches 32 rows with 2 x 16 bytes each.  scripts/micro/strided_read.hip (profiles/
// r3_node_measurements.txt) puts that pattern -- 128-byte row pieces, 32 rows per wavefront step -- at 2.3-2.7 TB/s whatever
// is in flight (the kernels: 2.4-2.5 TB/s; deeper prefetch and two-chunk bursts changed nothing, profiles/r5_mlp_bwd_*),
// against 4.2-5.5 TB/s for 256-byte pieces of a few rows per instruction.  Here a wavefront fetches its 32 rows x 64 columns
// as eight instructions of FOUR rows x 256 contiguous bytes, passes them through a wave-private padded LDS tile (row stride 68
// floats: both the 16-byte row-major writes and the fragment reads are conflict-free) and picks the A fragments up from
// there.  Work units are (128-row block) x (64-column super-chunk); the weight fragments keep their 32-column chunks (one
// workgroup barrier each).  Everything else -- running row exponent, split, epilogue, balanced ranges with float atomics
// for shared blocks -- as radial_mlp_bwd_pipe_kernel.
// Memory-queue order inside a super-chunk (vmcnt retires in order): the weight fragments of BOTH coming chunks are requested
// first, then the eight row pieces of the next super-chunk -- so neither hand-over of fragments to LDS waits for HBM.
// Preconditions (host): E > 0, W % 64 == 0, gridDim.x <= ceil(E / kMlpRows), g_emb zero-filled.
template <int H>
__global__ __launch_bounds__(256, 2) void radial_mlp_bwd_coal_kernel(const float* __restrict__ emb,
                                                                     const float* __restrict__ W0,
                                                                     const u32x4* __restrict__ Wb,
                                                                     const float* __restrict__ gw, float a0, int nb,
                                                                     int W, int64_t E, float* __restrict__ g_emb,
                                                                     const int* __restrict__ chunk_exp) {
  constexpr int NT = H / 32;
  constexpr int CH = 2 * 2 * NT * 64;  // uint4 per 32-column chunk of B fragments (two k-steps x two planes)
  constexpr int NV = CH / 256;
  constexpr int GS = H + 1;
  constexpr int kTS = 68;              // row stride of the transpose tile (floats)
  constexpr int kTileFloats = 32 * kTS;
  constexpr int kMainBytes = 2 * CH * 16 + 4 * kTileFloats * 4;
  constexpr int kEpiBytes = kMlpRows * GS * 4;
  constexpr int kBufBytes = kMainBytes > kEpiBytes ? kMainBytes : kEpiBytes;
  __shared__ __align__(16) unsigned char smem_raw[kBufBytes];
  __shared__ float w0s[H * kMaxNb];        // [k][c]
  __shared__ float w0t[kMaxNb * H];        // [c][k]
  __shared__ float es[kMlpRows * kMaxNb];  // embedding tile [row][c]
  __shared__ int rowexp[kMlpRows];
  u32x4* __restrict__ bsm = reinterpret_cast<u32x4*>(smem_raw);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const int l31 = lane & 31;
  float* tile = reinterpret_cast<float*>(smem_raw + 2 * CH * 16) + wv * kTileFloats;  // this wavefront's transpose tile
  const int nsc = W >> 6;  // 64-column super-chunks
  const int64_t nblk = (E + kMlpRows - 1) / kMlpRows;
  const int64_t U = nblk * nsc;
  const int64_t u0 = U * (int64_t)blockIdx.x / gridDim.x;
  const int64_t u1 = U * ((int64_t)blockIdx.x + 1) / gridDim.x;
  if (u0 >= u1) return;  // (workgroup-uniform)

  for (int i = tid; i < H * kMaxNb; i += 256) {
    const int k = i / kMaxNb, c = i - k * kMaxNb;
    const float v = c < nb ? W0[c * H + k] * a0 : 0.f;
    w0s[i] = v;
    w0t[c * H + k] = v;
  }
  constexpr int kUnset = 1 << 20;
  // transpose tile: this lane writes the float4 (row 4 q + lane / 16, columns 4 (lane % 16) ..) of load q and reads, as row
  // l31 of chunk hc, its 16 floats 32 hc + 16 half ..
  float* tw = tile + (lane >> 4) * kTS + 4 * (lane & 15);  // + 4 q kTS
  const float* tr = tile + l31 * kTS + 16 * half;          // + 32 hc + 8 s: the fragment order of Wb (k = 16 half + 8 s + t)

  auto rows_from_lanes = [&](int value, int (&out)[16]) __attribute__((always_inline)) {
    int* __restrict__ rb = rowexp + wv * 32;
    if (half == 0) rb[l31] = value;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int r = 0; r < 16; ++r) out[r] = rb[(r & 3) + 8 * (r >> 2) + 4 * half];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };

  int64_t u = u0;
  while (u < u1) {
    const int64_t blk = u / nsc;
    const int s0 = (int)(u - blk * nsc);
    const int n = (int)((u1 - u) < (int64_t)(nsc - s0) ? (u1 - u) : (int64_t)(nsc - s0));  // super-chunks of this segment
    const int s1 = s0 + n;
    u += n;
    const int c1 = 2 * s1;  // one past the segment's last 32-column chunk
    const int64_t blk0 = blk * kMlpRows;
    const float* __restrict__ gbase = gw + blk0 * W;  // wave-uniform
    const int rows_here = (int)(E - blk0 < (int64_t)kMlpRows ? E - blk0 : (int64_t)kMlpRows);
    // lane parts (bytes) of the eight row pieces: row 32 wv + 4 q + lane / 16 (clamped into a ragged block), 16 bytes at
    // 16 (lane % 16) of the super-chunk's 256
    unsigned goff[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int rl = wv * 32 + 4 * q + (lane >> 4);
      goff[q] = ((unsigned)(rl < rows_here ? rl : rows_here - 1) * (unsigned)W + 4u * (unsigned)(lane & 15)) * 4u;
    }

    __syncthreads();  // the previous segment's epilogue is done with smem_raw / es (first segment: w0s / w0t are in place)
    for (int i = tid; i < kMlpRows * kMaxNb; i += 256) {
      const int r = i / kMaxNb, c = i - r * kMaxNb;
      const float v = emb[(blk0 + (r < rows_here ? r : rows_here - 1)) * nb + (c < nb ? c : nb - 1)];
      es[i] = (c < nb && r < rows_here) ? v : 0.f;
    }

    mlp_v4f raw[8];  // (native vector type: the HIP float4 struct array went through scratch memory here)
    u32x4 pbA[NV], pbB[NV];
    auto load_rows = [&](int sc, mlp_v4f (&dst)[8]) __attribute__((always_inline)) {
      const int cc = sc < s1 ? sc : s1 - 1;  // past the segment: a valid super-chunk, never used
      const char* __restrict__ src = reinterpret_cast<const char*>(gbase + 64 * cc);
#pragma unroll
      for (int q = 0; q < 8; ++q) dst[q] = *reinterpret_cast<const mlp_v4f*>(src + goff[q]);
    };
    auto load_b = [&](int ch, u32x4 (&pb)[NV]) __attribute__((always_inline)) {
      const int cc = ch < c1 ? ch : c1 - 1;
      const u32x4* __restrict__ src = Wb + (int64_t)cc * CH;
#pragma unroll
      for (int v = 0; v < NV; ++v) pb[v] = src[tid + v * 256];
    };
    auto store_b = [&](int buf, const u32x4 (&pb)[NV]) __attribute__((always_inline)) {
#pragma unroll
      for (int v = 0; v < NV; ++v) bsm[buf * CH + tid + v * 256] = pb[v];
    };
    load_b(2 * s0, pbA);
    load_rows(s0, raw);
    store_b(0, pbA);
    __syncthreads();

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x16){0};
    int S = kUnset;                   // exponent of this lane's row (unset while the row is all zero)

    // one 32-column chunk: its 16 floats per lane come out of the transpose tile (k-steps 2 hc, 2 hc + 1 of the super-chunk)
    auto body = [&](int ch, int first, auto hc_tag, mlp_v4f (&rows)[8], u32x4 (&pb0)[NV], u32x4 (&pb1)[NV])
        __attribute__((always_inline)) {
      constexpr int hc = decltype(hc_tag)::value;  // 0 / 1: first / second chunk of the super-chunk; LDS weight buffer = hc
      float4 pa[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) pa[v] = *reinterpret_cast<const float4*>(tr + 32 * hc + 4 * v);
      const int we = chunk_exp[ch];
      float m = 0.f;  // the row's largest magnitude in this chunk (its other 16 values sit in lane ^ 32)
#pragma unroll
      for (int v = 0; v < 4; ++v)
        m = fmaxf(m, fmaxf(fmaxf(fabsf(pa[v].x), fabsf(pa[v].y)), fmaxf(fabsf(pa[v].z), fabsf(pa[v].w))));
      m = fmaxf(m, __shfl_xor(m, 32));
      int shift = 0;
      if (m > 0.f && m < 3.0e38f) {
        int em;
        (void)frexpf(m, &em);          // m < 2^em
        const int cap = 15 - em + we;  // largest S that keeps m 2^(S - we) below 2^15
        if (cap < S) {
          int ns = cap - 3;
          ns = ns > we + 100 ? we + 100 : (ns < we - 100 ? we - 100 : ns);
          shift = S == kUnset ? 0 : S - ns;
          S = ns;
        }
      }
      if (!first && __any(shift > 0)) {  // (rare after the first chunks: a new maximum 8x above every earlier one)
        int kr[16];
        rows_from_lanes(shift, kr);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t][r] = ldexpf(acc[t][r], -kr[r]);
      }
      int q = S == kUnset ? 0 : S - we;
      q = q > 120 ? 120 : (q < -120 ? -120 : q);
      const float qs = ldexpf(1.f, q);
      u32x4 ah[2], al[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        uint32_t a, b;
        split_pair_f16(pa[2 * s].x * qs, pa[2 * s].y * qs, a, b);
        ah[s][0] = a; al[s][0] = b;
        split_pair_f16(pa[2 * s].z * qs, pa[2 * s].w * qs, a, b);
        ah[s][1] = a; al[s][1] = b;
        split_pair_f16(pa[2 * s + 1].x * qs, pa[2 * s + 1].y * qs, a, b);
        ah[s][2] = a; al[s][2] = b;
        split_pair_f16(pa[2 * s + 1].z * qs, pa[2 * s + 1].w * qs, a, b);
        ah[s][3] = a; al[s][3] = b;
      }
      if constexpr (hc == 0) {
        // fragments of the two coming chunks, THEN the row pieces of the next super-chunk (see the header)
        load_b(ch + 1, pb0);
        load_b(ch + 2, pb1);
        load_rows((ch >> 1) + 1, rows);
      }
      const u32x4* __restrict__ bs = bsm + hc * CH + lane;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        u32x4 fb[2][NT];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int t = 0; t < NT; ++t) fb[p][t] = bs[((s * 2 + p) * NT + t) * 64];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = mfma_f16(al[s], fb[0][t], acc[t]);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = mfma_f16(ah[s], fb[1][t], acc[t]);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = mfma_f16(ah[s], fb[0][t], acc[t]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (hc == 0) store_b(1, pb0); else store_b(0, pb1);
      lds_barrier();
    };
    for (int j = 0; j < n; ++j) {
      // the super-chunk's rows: registers (requested one super-chunk ago) -> transpose tile
#pragma unroll
      for (int q = 0; q < 8; ++q) *reinterpret_cast<mlp_v4f*>(tw + 4 * q * kTS) = raw[q];
      body(2 * (s0 + j), j == 0, std::integral_constant<int, 0>{}, raw, pbA, pbB);
      body(2 * (s0 + j) + 1, 0, std::integral_constant<int, 1>{}, raw, pbA, pbB);
    }

    {  // accumulators back to the true scale: 2^-S of their row
      int sr[16];
      rows_from_lanes(S == kUnset ? 0 : S, sr);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = ldexpf(acc[t][r], -sr[r]);
    }
    // epilogue (exact fp32): as radial_mlp_bwd_pipe_kernel.  (every wavefront is past the last chunk's barrier: the weight
    // buffers and the transpose tiles are free)
    float* __restrict__ gp = reinterpret_cast<float*>(smem_raw);
    {
      const float* __restrict__ erow = es + (wv * 32 + l31) * kMaxNb;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        f32x16 pacc = {0};
#pragma unroll
        for (int s2 = 0; s2 < kMaxNb / 2; ++s2) {
          const float av = erow[2 * s2 + half];
          const float bv = w0t[(2 * s2 + half) * H + t * 32 + l31];
          pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, pacc, 0, 0, 0);
        }
        const int col = t * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int lr = wv * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          gp[lr * GS + col] = acc[t][r] * silu_grad_fast_f(pacc[r]);
        }
      }
    }
    __syncthreads();
    {
      const int r = tid >> 1, kh = tid & 1;
      float sacc[kMaxNb];
#pragma unroll
      for (int c = 0; c < kMaxNb; ++c) sacc[c] = 0.f;
      const float* __restrict__ gr = gp + r * GS + kh * (H / 2);
      const float* __restrict__ wr = w0s + kh * (H / 2) * kMaxNb;
#pragma unroll 4
      for (int k = 0; k < H / 2; ++k) {
        const float gv = gr[k];
        const float4 w0 = *reinterpret_cast<const float4*>(wr + k * kMaxNb);
        const float4 w1 = *reinterpret_cast<const float4*>(wr + k * kMaxNb + 4);
        sacc[0] += gv * w0.x; sacc[1] += gv * w0.y; sacc[2] += gv * w0.z; sacc[3] += gv * w0.w;
        sacc[4] += gv * w1.x; sacc[5] += gv * w1.y; sacc[6] += gv * w1.z; sacc[7] += gv * w1.w;
      }
#pragma unroll
      for (int c = 0; c < kMaxNb; ++c) sacc[c] += __shfl_xor(sacc[c], 1, 64);
      if (kh == 0 && r < rows_here) {
        float* __restrict__ dst = g_emb + (blk0 + r) * nb;
        if (n == nsc) {  // the whole block: plain stores (nobody else adds to these rows)
          for (int c = 0; c < nb; ++c) dst[c] = sacc[c];
        } else {
          for (int c = 0; c < nb; ++c) unsafeAtomicAdd(dst + c, sacc[c]);
        }
      }
    }
  }
}


// ---- f16x3 inference backward for NARROW outputs: all weight fragments resident in LDS -------------------------------------
// For W <= 256 (the first / last layer of the BASELINE models: W = 192, six 32-column chunks) the general kernels spend their
// time around the six chunks, not in them: with matrix instructions, weight staging and fragment reads all ablated the W = 192
// launch still takes 95 of its 96 us (profiles/r5_mlp_bwd_small.txt) -- per 128-row block a pipeline restart (first rows from
// HBM, first fragments from L2, two workgroup barriers), the embedding tile staged through LDS, and an epilogue that passes
// g_pre through a 66 KB LDS tile with two more barriers.  Here
//  * the launch is one 8-wavefront workgroup per CU that loads ALL fragments (W / 32 x 16 KiB) into LDS once; after that the
//    wavefronts are independent -- no workgroup barrier, no staging -- and walk contiguous ranges of 32-row blocks with the row
//    stream running two chunks ahead ACROSS block boundaries;
//  * the two operands of the matrix instruction are swapped (same fragments): the accumulators come out as
//    D[hidden][row] with lane = row, so the running exponent is the lane's own, the pre-activations are recomputed in the
//    same layout, and g_emb[row][:] = sum_hidden g_pre[hidden][row] W0s[:][hidden] is a dot product over the lane's own 64
//    registers (weights broadcast from a 4 KiB LDS table) + one exchange with lane ^ 32 -- no LDS tile, no barrier;
//  * every row block is whole (all chunks): plain stores, no zero-fill, no atomics.
// Preconditions (host): E > 0, W % 32 == 0, W / 32 * CH * 16 + 8.5 KiB of dynamic LDS granted, blockDim = 512.
template <int H>
__global__ __launch_bounds__(512, 2) void radial_mlp_bwd_small_kernel(const float* __restrict__ emb,
                                                                      const float* __restrict__ W0,
                                                                      const u32x4* __restrict__ Wb,
                                                                      const float* __restrict__ gw, float a0, int nb,
                                                                      int W, int64_t E, float* __restrict__ g_emb,
                                                                      const int* __restrict__ chunk_exp) {
  constexpr int NT = H / 32;
  constexpr int CH = 2 * 2 * NT * 64;  // uint4 per chunk of fragments (two k-steps x two planes)
  extern __shared__ __align__(16) unsigned char dyn_smem[];
  const int nchunks = W >> 5;
  u32x4* __restrict__ wsm = reinterpret_cast<u32x4*>(dyn_smem);                       // [nchunks][CH]
  float* __restrict__ w0q = reinterpret_cast<float*>(dyn_smem + (size_t)nchunks * CH * 16);  // [NT][16][2][kMaxNb]
  float* __restrict__ w0t = w0q + NT * 16 * 2 * kMaxNb;                                // [kMaxNb][H]
  int* __restrict__ cexp = reinterpret_cast<int*>(w0t + kMaxNb * H);                   // [nchunks]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const int l31 = lane & 31;

  for (int i = tid; i < nchunks * CH; i += 512) wsm[i] = Wb[i];
  for (int i = tid; i < NT * 16 * 2 * kMaxNb; i += 512) {
    const int c = i % kMaxNb, hf = (i / kMaxNb) & 1, r = (i / (2 * kMaxNb)) % 16, t = i / (32 * kMaxNb);
    const int k = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hf;  // hidden index of accumulator register r of tile t in lane-half hf
    w0q[i] = c < nb ? W0[c * H + k] * a0 : 0.f;
  }
  for (int i = tid; i < kMaxNb * H; i += 512) {
    const int c = i / H, k = i - c * H;
    w0t[i] = c < nb ? W0[c * H + k] * a0 : 0.f;
  }
  for (int i = tid; i < nchunks; i += 512) cexp[i] = chunk_exp[i];
  __syncthreads();

  // this wavefront's contiguous range of 32-row blocks
  const int64_t NB = (E + 31) / 32;
  const int64_t NW = (int64_t)gridDim.x * 8;
  const int64_t gwv = (int64_t)blockIdx.x * 8 + wv;
  const int64_t b0 = NB * gwv / NW, b1 = NB * (gwv + 1) / NW;
  if (b0 >= b1) return;
  const int64_t units = (b1 - b0) * nchunks;
  constexpr int kUnset = 1 << 20;

  // row stream: unit j = (block b0 + j / nchunks, chunk j % nchunks); this lane's 16 floats 32 ch + 16 half .. of its row
  auto row_ptr = [&](int64_t blk) {
    const int64_t row = blk * 32 + l31;
    return gw + (row < E ? row : E - 1) * W + 16 * half;
  };
  auto load_a = [&](const float* __restrict__ rp, int ch, float4 (&pa)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int v = 0; v < 4; ++v) pa[v] = *reinterpret_cast<const float4*>(rp + 32 * ch + 4 * v);
  };
  float4 paA[4], paB[4];
  // (position of the prefetch stream: two units ahead of the unit being consumed)
  int64_t pblk = b0;
  int pch = 0;
  const float* prow = row_ptr(pblk);
  auto advance = [&]() __attribute__((always_inline)) {
    if (++pch == nchunks) {
      pch = 0;
      pblk = pblk + 1 < b1 ? pblk + 1 : pblk;  // past the range: the last block again (valid, never used)
      prow = row_ptr(pblk);
    }
  };
  load_a(prow, pch, paA);
  advance();
  load_a(prow, pch, paB);
  advance();

  f32x16 acc[NT];
  int S = kUnset;
  int64_t blk = b0;
  int ch = 0;
  float ev[kMaxNb];
  auto load_ev = [&](int64_t b) __attribute__((always_inline)) {
    const int64_t row = b * 32 + l31;
    const float* __restrict__ er = emb + (row < E ? row : E - 1) * nb;
#pragma unroll
    for (int c = 0; c < kMaxNb; ++c) ev[c] = er[c < nb ? c : nb - 1];
#pragma unroll
    for (int c = 0; c < kMaxNb; ++c) ev[c] = c < nb ? ev[c] : 0.f;
  };

  auto body = [&](float4 (&pa)[4]) __attribute__((always_inline)) {
    if (ch == 0) {  // a new row block
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = (f32x16){0};
      S = kUnset;
      load_ev(blk);
    }
    const int we = cexp[ch];
    float m = 0.f;  // the row's largest magnitude in this chunk (its other 16 values sit in lane ^ 32)
#pragma unroll
    for (int v = 0; v < 4; ++v)
      m = fmaxf(m, fmaxf(fmaxf(fabsf(pa[v].x), fabsf(pa[v].y)), fmaxf(fabsf(pa[v].z), fabsf(pa[v].w))));
    m = fmaxf(m, __shfl_xor(m, 32));
    int shift = 0;
    if (m > 0.f && m < 3.0e38f) {
      int em;
      (void)frexpf(m, &em);          // m < 2^em
      const int cap = 15 - em + we;  // largest S that keeps m 2^(S - we) below 2^15
      if (cap < S) {
        int ns = cap - 3;
        ns = ns > we + 100 ? we + 100 : (ns < we - 100 ? we - 100 : ns);
        shift = S == kUnset ? 0 : S - ns;
        S = ns;
      }
    }
    if (ch > 0 && __any(shift > 0)) {  // (rare) the row is the lane's own: no exchange
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = ldexpf(acc[t][r], -shift);
    }
    int q = S == kUnset ? 0 : S - we;
    q = q > 120 ? 120 : (q < -120 ? -120 : q);
    const float qs = ldexpf(1.f, q);
    u32x4 ah[2], al[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      uint32_t a, b;
      split_pair_f16(pa[2 * s].x * qs, pa[2 * s].y * qs, a, b);
      ah[s][0] = a; al[s][0] = b;
      split_pair_f16(pa[2 * s].z * qs, pa[2 * s].w * qs, a, b);
      ah[s][1] = a; al[s][1] = b;
      split_pair_f16(pa[2 * s + 1].x * qs, pa[2 * s + 1].y * qs, a, b);
      ah[s][2] = a; al[s][2] = b;
      split_pair_f16(pa[2 * s + 1].z * qs, pa[2 * s + 1].w * qs, a, b);
      ah[s][3] = a; al[s][3] = b;
    }
    load_a(prow, pch, pa);  // two units ahead (this register set is free again)
    advance();
    const u32x4* __restrict__ bs = wsm + (int64_t)ch * CH + lane;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      u32x4 fb[2][NT];
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int t = 0; t < NT; ++t) fb[p][t] = bs[((s * 2 + p) * NT + t) * 64];
      __builtin_amdgcn_sched_barrier(0);
      // operands swapped: D[hidden][row], lane = row
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = mfma_f16(fb[0][t], al[s], acc[t]);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = mfma_f16(fb[1][t], ah[s], acc[t]);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = mfma_f16(fb[0][t], ah[s], acc[t]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (++ch < nchunks) return;
    // ---- the block is complete: epilogue in registers
    ch = 0;
    const float back = ldexpf(1.f, S == kUnset ? 0 : -S);  // accumulators back to the true scale (|S| <= ~130: finite)
    float sacc[kMaxNb];
#pragma unroll
    for (int c = 0; c < kMaxNb; ++c) sacc[c] = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f32x16 pacc = {0};
#pragma unroll
      for (int s2 = 0; s2 < kMaxNb / 2; ++s2) {
        const float av = w0t[(2 * s2 + half) * H + t * 32 + l31];
        const float bv = half ? ev[2 * s2 + 1] : ev[2 * s2];
        pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, pacc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float gp = acc[t][r] * back * silu_grad_fast_f(pacc[r]);
        const float4 wa = *reinterpret_cast<const float4*>(w0q + ((t * 16 + r) * 2 + half) * kMaxNb);
        const float4 wb = *reinterpret_cast<const float4*>(w0q + ((t * 16 + r) * 2 + half) * kMaxNb + 4);
        sacc[0] += gp * wa.x; sacc[1] += gp * wa.y; sacc[2] += gp * wa.z; sacc[3] += gp * wa.w;
        sacc[4] += gp * wb.x; sacc[5] += gp * wb.y; sacc[6] += gp * wb.z; sacc[7] += gp * wb.w;
      }
    }
#pragma unroll
    for (int c = 0; c < kMaxNb; ++c) sacc[c] += __shfl_xor(sacc[c], 32, 64);
    const int64_t row = blk * 32 + l31;
    if (half == 0 && row < E) {
      float* __restrict__ dst = g_emb + row * nb;
      for (int c = 0; c < nb; ++c) dst[c] = sacc[c];
    }
    ++blk;
  };
  for (int64_t j = 0; j < units; j += 2) {
    body(paA);
    if (j + 1 < units) body(paB);
  }
}

}  // namespace nqa
