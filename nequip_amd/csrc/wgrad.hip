// Parameter gradients of the dense maps on the path (training only): the reductions over edges / atoms that autograd
// produces for
//   ScalarMLPFunction layers   dW[i, j]    = sum_e a[e, i] g[e, j]                  (nequip/nn/mlp.py:262-268)
//   e3nn o3.Linear             dW[u, w]    = sum_z sum_m x[z, u, m] g[z, w, m]      (interaction_block.py:82-87,129-138)
//   self-connection FCTP       dW[t, u, w] = sum_{z: type z = t} sum_m x[z,u,m] g[z,w,m]  (interaction_block.py:142-146)
// i.e. C = A^T B with a tiny output (<= 128 x 704) and a reduction length of 10^4..10^6 rows.  Library GEMMs run these
// shapes on a handful of workgroups (no split over the reduction): 0.08-0.5 ms each, 7 ms of a 21 ms cfg-4 training
// step.  Here the reduction is split over S row ranges; one wavefront owns a (<=128) x 64 output tile of one range and
// feeds fp32 MFMA 32x32x2 straight from global memory -- both operands are read in their natural row-major layout
// (lane = output row / column, the two k-slots of the instruction = two consecutive rows), so there is no transposition
// and no LDS in the main loop.  The four wavefronts of a workgroup own four consecutive row ranges of the SAME output
// tile and add their accumulators through LDS (fixed order) before one of them stores the tile: S partial tiles go to a
// workspace [S][n_types][weight_numel] (a quarter of the wavefronts' count), the caller sums over S (deterministic: no
// atomics).  Operands of the next step are loaded before the MFMAs of the current one.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>

#include "plan.h"

namespace nqa {

struct WgradInstr {  // one weight matrix [M, N] (row-major at out_off); A block at a_off, B block at b_off, 2l+1 = d
  int32_t a_off, b_off, M, N, d, out_off, unit_begin, mt;
};
static_assert(sizeof(WgradInstr) == 32, "WgradInstr layout");

constexpr int kMaxWgradInstr = 64;
constexpr int kWgU = 4;  // row pairs per pipeline step (8 rows)

struct WgradArgs {
  const float* __restrict__ A;        // [Z, lda]: element (z, i, m) of an instruction at a_off + i*d + m
  const float* __restrict__ B;        // [Z, ldb]
  const int64_t* __restrict__ types;  // optional [Z]
  float* __restrict__ partials;       // [S][T][out_stride]
  int64_t lda, ldb, Z, out_stride;
  int32_t T, S, n_instr, zc;  // zc: rows per split (multiple of 2*kWgU)
  int32_t total_units, pad;
  WgradInstr instr[kMaxWgradInstr];
};

typedef float wg_f16 __attribute__((ext_vector_type(16)));

// RB: 32-row blocks of the output tile owned by a wavefront (M <= 32*RB per tile), two 32-column blocks.
// WGRED: the four wavefronts of a workgroup share one (tile, partial) unit (reduced on chip); otherwise one unit each
template <int RB, bool TYPED, bool WGRED>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradArgs a) {
  const int lane = threadIdx.x & 63, half = lane >> 5, li = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int unit = WGRED ? (int)blockIdx.x : (int)blockIdx.x * 4 + wave;
  if (!WGRED && unit >= a.total_units) return;
  int qi = 0;
  while (qi + 1 < a.n_instr && a.instr[qi + 1].unit_begin <= unit) ++qi;
  const WgradInstr q = a.instr[qi];
  const int nt = (q.N + 63) >> 6;
  int local = unit - q.unit_begin;
  const int nt_i = local % nt;
  local /= nt;
  const int mt_i = local % q.mt;
  local /= q.mt;
  const int s = local % a.S;
  const int t = local / a.S;
  const int m0 = mt_i * 32 * RB, n0 = nt_i * 64;
  const int64_t z_begin0 = (WGRED ? (int64_t)s * 4 + wave : (int64_t)s) * a.zc;  // zc: rows per wavefront
  const int64_t z_begin = z_begin0 < a.Z ? z_begin0 : a.Z;
  const int64_t z_end = z_begin + a.zc < a.Z ? z_begin + a.zc : a.Z;

  // addressing: wave-uniform base of the step's first row (scalar registers) + a 32-bit per-lane byte offset
  uint32_t ao[RB], bo[2], amask[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int i = m0 + rb * 32 + li;
    amask[rb] = i < q.M ? 0xFFFFFFFFu : 0u;
    ao[rb] = 4u * (uint32_t)(q.a_off + (i < q.M ? i : q.M - 1) * q.d);
  }
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const int j = n0 + cb * 32 + li;
    bo[cb] = 4u * (uint32_t)(q.b_off + (j < q.N ? j : q.N - 1) * q.d);
  }
  const uint32_t lda4 = 4u * (uint32_t)a.lda, ldb4 = 4u * (uint32_t)a.ldb;

  wg_f16 acc[RB][2];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][cb][r] = 0.f;

  float av[2][RB][kWgU], bv[2][2][kWgU];
  uint32_t msk[2][kWgU];
  // operands of pipeline step (zb, m): rows zb + 2p + half, p < kWgU.  Every load is unconditional (addresses are
  // clamped into the range); rows outside the range / of another atom type are zeroed through a bit mask on the A
  // operand that is applied when the value is consumed, so nothing waits on the loads inside the request phase.
  auto load = [&](int buf, int64_t zb, int m, bool live) {
    const char* __restrict__ as = reinterpret_cast<const char*>(a.A + zb * a.lda + m);  // uniform
    const char* __restrict__ bs = reinterpret_cast<const char*>(a.B + zb * a.ldb + m);
#pragma unroll
    for (int p = 0; p < kWgU; ++p) {
      const bool in = live && (zb + 2 * p + half < z_end);
      const uint32_t row = in ? (uint32_t)(2 * p + half) : 0u;  // row zb itself is always inside the range
      uint32_t mk = in ? 0xFFFFFFFFu : 0u;
      if (TYPED) {
        const int64_t ty = *reinterpret_cast<const int64_t*>(reinterpret_cast<const char*>(a.types + zb) + 8u * row);
        msk[buf][p] = ty == (int64_t)t ? mk : 0u;
      } else {
        msk[buf][p] = mk;
      }
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) av[buf][rb][p] = *reinterpret_cast<const float*>(as + (row * lda4 + ao[rb]));
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) bv[buf][cb][p] = *reinterpret_cast<const float*>(bs + (row * ldb4 + bo[cb]));
    }
  };
  auto mfma_all = [&](int buf) {
#pragma unroll
    for (int p = 0; p < kWgU; ++p)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const float x =
            __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, av[buf][rb][p]) & msk[buf][p] & amask[rb]);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
          acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, bv[buf][cb][p], acc[rb][cb], 0, 0, 0);
      }
  };
  if (z_begin < z_end) {
    // steps (row block zb, component m), m fastest; the count is padded to an even number (the padding step re-reads the
    // last block with a zero A operand) so that the two-phase loop has a single exit and the accumulators stay in place
    const int64_t nzb = (z_end - z_begin + 2 * kWgU - 1) / (2 * kWgU);
    const int64_t nsteps = nzb * q.d;
    int64_t zb = z_begin;
    int m = 0;
    int64_t it = 0;
    // one pipeline phase: request the operands of the next step into the other register set, then issue the MFMAs
    // of the current one (buffer indices are literals at both call sites: no dynamic register indexing)
    auto phase = [&](int cur_buf, int nxt_buf) {
      int64_t zb_n = zb;
      int m_n = m + 1;
      if (m_n == q.d) {
        m_n = 0;
        zb_n = zb + 2 * kWgU;
      }
      ++it;
      const bool live = it < nsteps;
      if (!live) {
        zb_n = zb;
        m_n = 0;
      }
      load(nxt_buf, zb_n, m_n, live);
      __builtin_amdgcn_sched_barrier(0);
      mfma_all(cur_buf);
      __builtin_amdgcn_sched_barrier(0);
      zb = zb_n;
      m = m_n;
    };
    load(0, zb, m, true);
    const int64_t npairs = (nsteps + 1) / 2;
    for (int64_t pr = 0; pr < npairs; ++pr) {
      phase(0, 1);
      phase(1, 0);
    }
  }
  // workgroup reduction in a fixed order: wavefronts 1..3 hand their accumulators to wavefront 0 through LDS
  __shared__ float red[WGRED ? RB * 2 * 16 * 64 : 1];
#pragma unroll 1
  for (int src = 1; WGRED && src < 4; ++src) {
    if (wave == src) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[((rb * 2 + cb) * 16 + r) * 64 + lane] = acc[rb][cb][r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[rb][cb][r] += red[((rb * 2 + cb) * 16 + r) * 64 + lane];
    }
    __syncthreads();
  }
  if (WGRED && wave != 0) return;
  float* outp = a.partials + ((int64_t)s * a.T + t) * a.out_stride + q.out_off;
  const bool full = (m0 + 32 * RB <= q.M) && (n0 + 64 <= q.N);  // wave-uniform: interior tiles store unpredicated
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int j = n0 + cb * 32 + li;
      if (full) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = m0 + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          outp[(int64_t)i * q.N + j] = acc[rb][cb][r];
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = m0 + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (i < q.M && j < q.N) outp[(int64_t)i * q.N + j] = acc[rb][cb][r];
        }
      }
    }
}

// ---- split-bf16 form (output tiles wider than 64 rows: the radial-MLP weight gradients) ---------------------------------
// The same reduction on v_mfma_f32_32x32x16_bf16: a pipeline step is 16 rows, lane half kg owns rows zb + 8 kg + t
// (t = 0..7) of its A column / B column, both operands are split into three bf16 terms in registers (six partial products,
// fp32-accurate: as radial_mlp.hip) -- 24 MFMAs x 32 cycles per 16 rows and 64 x 64 tile instead of 32 x 64 on the fp32
// matrix path.  One wavefront owns a 64 x 64 tile (two row blocks: the raw operands of the next step, 80 registers, stay
// in flight during the MFMAs of the current one).  Measured: the 128-row calls of the training step 0.95 -> 0.83 ms only --
// 32 scalar row loads (8 KB) per step and wavefront are 80 B/clk per CU at the MFMA rate, more than the 64 B/clk the
// vector memory path delivers; a wider tile per wavefront (or LDS staging shared by the workgroup) is what it needs next.
typedef __attribute__((ext_vector_type(8))) __bf16 wg_bf16x8;
typedef __attribute__((ext_vector_type(4))) uint32_t wg_u32x4;

__device__ __forceinline__ uint32_t wg_cvt_pk_bf16(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
__device__ __forceinline__ void wg_split8(const float (&v)[8], wg_u32x4& h, wg_u32x4& m, wg_u32x4& l) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float x0 = v[2 * p], x1 = v[2 * p + 1];
    const uint32_t hh = wg_cvt_pk_bf16(x0, x1);
    float r0 = x0 - __uint_as_float(hh << 16);
    float r1 = x1 - __uint_as_float(hh & 0xffff0000u);
    const uint32_t mm = wg_cvt_pk_bf16(r0, r1);
    r0 -= __uint_as_float(mm << 16);
    r1 -= __uint_as_float(mm & 0xffff0000u);
    h[p] = hh;
    m[p] = mm;
    l[p] = wg_cvt_pk_bf16(r0, r1);
  }
}
__device__ __forceinline__ wg_f16 wg_mfma_bf16(const wg_u32x4& a, const wg_u32x4& b, const wg_f16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wg_bf16x8, a), __builtin_bit_cast(wg_bf16x8, b), c, 0,
                                                 0, 0);
}

constexpr int kWgRows = 16;  // rows per pipeline step of the split form

// (174 registers; asking for three wavefronts per SIMD spills 36 and doubles the kernel's time: measured)
template <bool TYPED, bool WGRED>
__global__ __launch_bounds__(256, 2) void wgrad_split_kernel(const WgradArgs a) {
  constexpr int RB = 2;
  const int lane = threadIdx.x & 63, half = lane >> 5, li = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int unit = WGRED ? (int)blockIdx.x : (int)blockIdx.x * 4 + wave;
  if (!WGRED && unit >= a.total_units) return;
  int qi = 0;
  while (qi + 1 < a.n_instr && a.instr[qi + 1].unit_begin <= unit) ++qi;
  const WgradInstr q = a.instr[qi];
  const int nt = (q.N + 63) >> 6;
  int local = unit - q.unit_begin;
  const int nt_i = local % nt;
  local /= nt;
  const int mt_i = local % q.mt;
  local /= q.mt;
  const int s = local % a.S;
  const int t = local / a.S;
  const int m0 = mt_i * 32 * RB, n0 = nt_i * 64;
  const int64_t z_begin0 = (WGRED ? (int64_t)s * 4 + wave : (int64_t)s) * a.zc;
  const int64_t z_begin = z_begin0 < a.Z ? z_begin0 : a.Z;
  const int64_t z_end = z_begin + a.zc < a.Z ? z_begin + a.zc : a.Z;

  uint32_t ao[RB], bo[2], amask[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int i = m0 + rb * 32 + li;
    amask[rb] = i < q.M ? 0xFFFFFFFFu : 0u;
    ao[rb] = 4u * (uint32_t)(q.a_off + (i < q.M ? i : q.M - 1) * q.d);
  }
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const int j = n0 + cb * 32 + li;
    bo[cb] = 4u * (uint32_t)(q.b_off + (j < q.N ? j : q.N - 1) * q.d);
  }
  const uint32_t lda4 = 4u * (uint32_t)a.lda, ldb4 = 4u * (uint32_t)a.ldb;

  wg_f16 acc[RB][2];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][cb][r] = 0.f;

  float av[2][RB][8], bv[2][2][8];
  uint32_t msk[2][8];
  // operands of step (zb, m): rows zb + 8 half + t.  Loads are unconditional (rows clamped to zb, which is inside the
  // range); rows outside the range / of another atom type are zeroed through the mask when the A values are consumed.
  auto load = [&](int buf, int64_t zb, int m, bool live) __attribute__((always_inline)) {
    const char* __restrict__ as = reinterpret_cast<const char*>(a.A + zb * a.lda + m);  // uniform
    const char* __restrict__ bs = reinterpret_cast<const char*>(a.B + zb * a.ldb + m);
#pragma unroll
    for (int tt = 0; tt < 8; ++tt) {
      const bool in = live && (zb + 8 * half + tt < z_end);
      const uint32_t row = in ? (uint32_t)(8 * half + tt) : 0u;
      const uint32_t mk = in ? 0xFFFFFFFFu : 0u;
      if (TYPED) {
        const int64_t ty = *reinterpret_cast<const int64_t*>(reinterpret_cast<const char*>(a.types + zb) + 8u * row);
        msk[buf][tt] = ty == (int64_t)t ? mk : 0u;
      } else {
        msk[buf][tt] = mk;
      }
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) av[buf][rb][tt] = *reinterpret_cast<const float*>(as + (row * lda4 + ao[rb]));
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) bv[buf][cb][tt] = *reinterpret_cast<const float*>(bs + (row * ldb4 + bo[cb]));
    }
  };
  auto mfma_all = [&](int buf) __attribute__((always_inline)) {
    wg_u32x4 bh[2], bm[2], bl[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) wg_split8(bv[buf][cb], bh[cb], bm[cb], bl[cb]);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      float x[8];
#pragma unroll
      for (int tt = 0; tt < 8; ++tt)
        x[tt] = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, av[buf][rb][tt]) & msk[buf][tt] & amask[rb]);
      wg_u32x4 ah, am, al;
      wg_split8(x, ah, am, al);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        acc[rb][cb] = wg_mfma_bf16(am, bm[cb], acc[rb][cb]);
        acc[rb][cb] = wg_mfma_bf16(ah, bl[cb], acc[rb][cb]);
        acc[rb][cb] = wg_mfma_bf16(al, bh[cb], acc[rb][cb]);
        acc[rb][cb] = wg_mfma_bf16(ah, bm[cb], acc[rb][cb]);
        acc[rb][cb] = wg_mfma_bf16(am, bh[cb], acc[rb][cb]);
        acc[rb][cb] = wg_mfma_bf16(ah, bh[cb], acc[rb][cb]);
      }
    }
  };
  if (z_begin < z_end) {
    const int64_t nzb = (z_end - z_begin + kWgRows - 1) / kWgRows;
    const int64_t nsteps = nzb * q.d;
    int64_t zb = z_begin;
    int m = 0;
    int64_t it = 0;
    auto phase = [&](int cur_buf, int nxt_buf) __attribute__((always_inline)) {
      int64_t zb_n = zb;
      int m_n = m + 1;
      if (m_n == q.d) {
        m_n = 0;
        zb_n = zb + kWgRows;
      }
      ++it;
      const bool live = it < nsteps;
      if (!live) {
        zb_n = zb;
        m_n = 0;
      }
      load(nxt_buf, zb_n, m_n, live);
      __builtin_amdgcn_sched_barrier(0);
      mfma_all(cur_buf);
      __builtin_amdgcn_sched_barrier(0);
      zb = zb_n;
      m = m_n;
    };
    load(0, zb, m, true);
    const int64_t npairs = (nsteps + 1) / 2;
    for (int64_t pr = 0; pr < npairs; ++pr) {
      phase(0, 1);
      phase(1, 0);
    }
  }
  __shared__ float red[WGRED ? RB * 2 * 16 * 64 : 1];
#pragma unroll 1
  for (int src = 1; WGRED && src < 4; ++src) {
    if (wave == src) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[((rb * 2 + cb) * 16 + r) * 64 + lane] = acc[rb][cb][r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[rb][cb][r] += red[((rb * 2 + cb) * 16 + r) * 64 + lane];
    }
    __syncthreads();
  }
  if (WGRED && wave != 0) return;
  float* outp = a.partials + ((int64_t)s * a.T + t) * a.out_stride + q.out_off;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int j = n0 + cb * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = m0 + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (i < q.M && j < q.N) outp[(int64_t)i * q.N + j] = acc[rb][cb][r];
      }
    }
}

}  // namespace nqa


using namespace nqa;

extern "C" {

// NQA_WGRAD_EXACT_FP32=1: every shape on the fp32 matrix path (no split-bf16 kernel)
static bool wgrad_split_enabled() {
  static const bool on = [] {
    const char* e = std::getenv("NQA_WGRAD_EXACT_FP32");
    return !(e && e[0] != '\0' && e[0] != '0');
  }();
  return on;
}
// rows-of-32 per output tile: 1 / 2 for M <= 32 / 64; wider outputs take 64-row tiles of the split-bf16 kernel (or
// 128-row tiles of the fp32 kernel when that is switched off)
static int wgrad_rb(int maxM, bool* split) {
  *split = maxM > 64 && wgrad_split_enabled();
  return maxM <= 32 ? 1 : ((maxM <= 64 || *split) ? 2 : 4);
}

static bool wgrad_wg_reduce() {
  static const bool on = [] {
    const char* e = std::getenv("NQA_WGRAD_WG_REDUCE");
    return !(e && e[0] == '0');
  }();
  return on;
}

int32_t nqa_wgrad_splits(const void* instr_table, int32_t n_instr, int32_t n_types, int64_t num_rows) {
  if (!instr_table || n_instr <= 0 || n_instr > kMaxWgradInstr || n_types < 1 || num_rows < 0) return NQA_ERR_INVALID;
  const int32_t* tab = static_cast<const int32_t*>(instr_table);
  int maxM = 0;
  for (int i = 0; i < n_instr; ++i) maxM = tab[6 * i + 2] > maxM ? tab[6 * i + 2] : maxM;
  bool split = false;
  const int RB = wgrad_rb(maxM, &split);
  int64_t tiles = 0;
  for (int i = 0; i < n_instr; ++i) {
    const int M = tab[6 * i + 2], N = tab[6 * i + 3];
    tiles += (int64_t)((M + 32 * RB - 1) / (32 * RB)) * ((N + 63) / 64);
  }
  tiles *= n_types;
  if (tiles <= 0) return NQA_ERR_INVALID;
  // ~2 wavefronts per SIMD on 256 CUs (four wavefronts = one workgroup per partial), at least 64 rows per wavefront
  const int wpu = wgrad_wg_reduce() ? 4 : 1;  // wavefronts per partial
  int64_t S = (2048 / wpu + tiles - 1) / tiles;
  const int64_t max_s = (num_rows + 64 * wpu - 1) / (64 * wpu);
  if (S > max_s) S = max_s;
  if (S < 1) S = 1;
  return (int32_t)S;
}

int nqa_wgrad(int32_t dtype, const void* a_rows, const void* b_rows, const int64_t* row_types,
              const void* instr_table, int32_t n_instr, int64_t lda, int64_t ldb, int64_t num_rows, int32_t n_types,
              int64_t out_stride, int32_t splits, void* partials, nqa_stream stream) {
  if (dtype != NQA_F32) {
    set_error("nqa_wgrad: float32 only");
    return NQA_ERR_UNSUPPORTED;
  }
  if (!instr_table || n_instr <= 0 || n_instr > kMaxWgradInstr || n_types < 1 || (n_types > 1 && !row_types) ||
      num_rows < 0 || splits < 1 || out_stride <= 0 || lda <= 0 || ldb <= 0 || lda > (1 << 24) || ldb > (1 << 24) || !partials || (num_rows > 0 && (!a_rows || !b_rows))) {
    set_error("nqa_wgrad: invalid argument");
    return NQA_ERR_INVALID;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (num_rows == 0) {
    if (hipMemsetAsync(partials, 0, sizeof(float) * (size_t)splits * n_types * out_stride, s) != hipSuccess) {
      set_error("nqa_wgrad: memset failed");
      return NQA_ERR_LAUNCH;
    }
    return NQA_OK;
  }
  WgradArgs a{};
  a.A = static_cast<const float*>(a_rows);
  a.B = static_cast<const float*>(b_rows);
  a.types = n_types > 1 ? row_types : nullptr;
  a.partials = static_cast<float*>(partials);
  a.lda = lda;
  a.ldb = ldb;
  a.Z = num_rows;
  a.out_stride = out_stride;
  a.T = n_types;
  a.S = splits;
  a.n_instr = n_instr;
  const bool wgred = wgrad_wg_reduce();
  const int64_t nranges = (wgred ? 4 : 1) * (int64_t)splits;
  int64_t zc = (num_rows + nranges - 1) / nranges;  // rows per wavefront
  zc = (zc + kWgRows - 1) / kWgRows * kWgRows;  // (16: a multiple of both kernels' step)
  if (zc > 2147483647LL / 2) {
    set_error("nqa_wgrad: split too long");
    return NQA_ERR_UNSUPPORTED;
  }
  a.zc = (int32_t)zc;
  const int32_t* tab = static_cast<const int32_t*>(instr_table);
  int maxM = 0;
  for (int i = 0; i < n_instr; ++i) maxM = tab[6 * i + 2] > maxM ? tab[6 * i + 2] : maxM;
  bool split = false;
  const int RB = wgrad_rb(maxM, &split);
  int64_t units = 0;
  for (int i = 0; i < n_instr; ++i) {
    WgradInstr& q = a.instr[i];
    q.a_off = tab[6 * i + 0];
    q.b_off = tab[6 * i + 1];
    q.M = tab[6 * i + 2];
    q.N = tab[6 * i + 3];
    q.d = tab[6 * i + 4];
    q.out_off = tab[6 * i + 5];
    if (q.M <= 0 || q.N <= 0 || q.d <= 0 || q.a_off < 0 || q.b_off < 0 || q.out_off < 0 ||
        (int64_t)q.out_off + (int64_t)q.M * q.N > out_stride || (int64_t)q.a_off + (int64_t)q.M * q.d > lda ||
        (int64_t)q.b_off + (int64_t)q.N * q.d > ldb) {
      set_error("nqa_wgrad: instruction outside its operand rows / the weight vector");
      return NQA_ERR_INVALID;
    }
    q.mt = (q.M + 32 * RB - 1) / (32 * RB);
    q.unit_begin = (int32_t)units;
    units += (int64_t)n_types * splits * q.mt * ((q.N + 63) / 64);
    if (units > 2147483647LL) {
      set_error("nqa_wgrad: too many work units");
      return NQA_ERR_UNSUPPORTED;
    }
  }
  a.total_units = (int32_t)units;
  const dim3 grid((unsigned)(wgred ? units : (units + 3) / 4));
#define NQA_WGRAD_LAUNCH(R)                                                        \
  if (a.types != nullptr) {                                                        \
    if (wgred) hipLaunchKernelGGL((wgrad_kernel<R, true, true>), grid, dim3(256), 0, s, a);    \
    else hipLaunchKernelGGL((wgrad_kernel<R, true, false>), grid, dim3(256), 0, s, a);         \
  } else {                                                                         \
    if (wgred) hipLaunchKernelGGL((wgrad_kernel<R, false, true>), grid, dim3(256), 0, s, a);   \
    else hipLaunchKernelGGL((wgrad_kernel<R, false, false>), grid, dim3(256), 0, s, a);        \
  }
  if (split) {
    if (a.types != nullptr) {
      if (wgred) hipLaunchKernelGGL((wgrad_split_kernel<true, true>), grid, dim3(256), 0, s, a);
      else hipLaunchKernelGGL((wgrad_split_kernel<true, false>), grid, dim3(256), 0, s, a);
    } else {
      if (wgred) hipLaunchKernelGGL((wgrad_split_kernel<false, true>), grid, dim3(256), 0, s, a);
      else hipLaunchKernelGGL((wgrad_split_kernel<false, false>), grid, dim3(256), 0, s, a);
    }
  } else if (RB == 1) {
    NQA_WGRAD_LAUNCH(1)
  } else if (RB == 2) {
    NQA_WGRAD_LAUNCH(2)
  } else {
    NQA_WGRAD_LAUNCH(4)
  }
#undef NQA_WGRAD_LAUNCH
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_error(std::string("nqa_wgrad: ") + hipGetErrorString(err));
    return NQA_ERR_LAUNCH;
  }
  return NQA_OK;
}

}  // extern "C"
