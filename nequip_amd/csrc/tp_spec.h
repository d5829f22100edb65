// Internal: interface between the generated structure-specialised kernels (gen_spec.py) and the dispatcher.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <string>

namespace nqa {

template <typename T>
struct SpecArgs {
  const T* __restrict__ x;   // [N, din]        (fwd, bwd_edge)
  const T* __restrict__ x2;  // [N, din] / y2 [E, S]: second operand set of the dual pair kernel (which = 6)
  const T* __restrict__ y2;
  const T* __restrict__ w2;  // forward JVP (which = 7): cotangent of the weights, rows as w
  const T* __restrict__ y;   // [E, S]
  const T* __restrict__ w;   // [E, wn]
  const T* __restrict__ g;   // [N, dout]       (bwd_edge, bwd_x)
  T* __restrict__ out;       // fwd: [N, dout]; bwd_x: [N, din]
  T* __restrict__ gw;        // [E, wn] or null
  T* __restrict__ gy;        // [E, gy_stride] or null (gy itself when mul <= 64, else per-chunk partials)
  T* __restrict__ gxe;       // fused backward: per-edge grad_x contributions [E, xd, mul] (bwd_edge writes, sum reads)
  const int32_t* __restrict__ rowptr;
  const int32_t* __restrict__ eid;
  const int32_t* __restrict__ nbr;
  const int32_t* __restrict__ eid2;  // pair-centric backward: the pair's second directed edge (owner -> other)
  // Weight rows (paired radial weights, nqa_tp_scatter_*_paired): wid[slot] in CSR slot order is the row of grad_w the
  // edge writes, row (wid < wP ? wid : wid - wP) of w holds its weights.  Unpaired calls pass wid = eid, wP = INT32_MAX.
  const int32_t* __restrict__ wid;
  int32_t wP;
  int32_t N;
  int32_t mul;
  int32_t din, dout, wn;
  int32_t gy_stride;
  int32_t gx_atomic;  // which = 4: the other node's grad_x goes into the zeroed accumulator gxe [N, din] by atomics (ring kernel)
  int32_t gy_atomic;  // which = 4, ring kernels, more than one (chunk, part) per edge: gy is the zeroed grad_y [E, S] itself and
                      // every wavefront adds its sums to it (no partial rows, no reduce pass)
};

// which: 0 = fwd, 1 = bwd_edge (+ gxe rows when a.gxe != null), 2 = bwd_x, 3 = per-source sum of the gxe rows,
// 4 = pair-centric backward (owner CSR), 5 = out += per-node sum of the pair rows, 6 = dual pair-centric edge gradients, 7 = forward JVP, 8 = dual bwd_x,
// 9 = out += accumulator rows (a.gx_atomic form of 4);  wpn: requested wavefronts per (node, chunk)
using SpecLaunchFn = int (*)(int which, int wpn, const SpecArgs<float>& a, hipStream_t stream);

struct SpecEntry {
  std::string key;
  SpecLaunchFn launch;
  int xd, s, od, np;
  int pair;  // pair-centric backward (which = 4 / 5): 0 not generated, 1 one wavefront per (node, chunk), n > 1 split
             // over n wavefronts by input block (grad_y partials: nchunk * n per edge)
  int ring;  // the accumulator (atomic) form of the pair kernel's grad_x and its last step (which = 9) exist for multiples of
             // 64 channels: 1 = in the LDS-ring kernel, 2 = in the split kernel; 0 = rows only
  SpecEntry* next;
};

SpecEntry*& spec_registry_head();
const SpecEntry* find_spec(const std::string& key);

struct SpecRegistrar {
  SpecEntry entry;
  SpecRegistrar(const char* key, SpecLaunchFn fn, int xd, int s, int od, int np, int pair, int ring = 0) {
    entry.key = key;
    entry.launch = fn;
    entry.xd = xd;
    entry.s = s;
    entry.od = od;
    entry.np = np;
    entry.pair = pair;
    entry.ring = ring;
    entry.next = spec_registry_head();
    spec_registry_head() = &entry;
  }
};

// Workgroup b is dispatched to XCD b % 8 (observed placement, used for speed only): hand each XCD a contiguous
// range of work items so that the node rows gathered by neighbouring atoms (x[src], grad_out[dst]) are re-used out
// of that XCD's private 4 MiB L2 instead of being fetched by all eight.  Bijective for any grid size.
__device__ __forceinline__ unsigned spec_xcd_remap(unsigned b, unsigned nblk) {
  const unsigned q = nblk >> 3, r = nblk & 7u;
  const unsigned xcd = b & 7u, idx = b >> 3;
  const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// Address = wave-uniform row base (scalar registers) + per-lane 32-bit byte offset: lets the compiler use the
// `global_load/store v, v_off, s[base:base+1] offset:imm` form (no 64-bit vector address arithmetic per access).
template <typename T>
__device__ __forceinline__ T* spec_at(T* uniform_base, unsigned lane_bytes) {
  return reinterpret_cast<T*>(reinterpret_cast<char*>(uniform_base) + lane_bytes);
}
template <typename T>
__device__ __forceinline__ const T* spec_at(const T* uniform_base, unsigned lane_bytes) {
  return reinterpret_cast<const T*>(reinterpret_cast<const char*>(uniform_base) + lane_bytes);
}
__device__ __forceinline__ int spec_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T>
__device__ __forceinline__ int spec_gwrow(const SpecArgs<T>& a, int slot) {
  return spec_uniform(a.wid[slot]);
}
template <typename T>
__device__ __forceinline__ int spec_wrow_of(const SpecArgs<T>& a, int gwrow) {
  return gwrow >= a.wP ? gwrow - a.wP : gwrow;
}
template <typename T>
__device__ __forceinline__ int spec_wrow(const SpecArgs<T>& a, int slot) {
  return spec_wrow_of(a, spec_gwrow(a, slot));
}

// More than 64 KiB of dynamic LDS needs the function attribute, per device: set at a kernel's first launch on each device of
// this process (host side).  `done`: the caller's per-kernel-instantiation table (a static at the launch site).
static inline bool spec_allow_lds(const void* fn, int bytes, bool (&done)[64]) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  if (!done[dev]) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
    done[dev] = true;
  }
  return true;
}

// ---- LDS-DMA (global -> LDS without registers) ------------------------------------------------------------------
// One instruction moves 64 lanes x 16 (or 4) bytes: lane l's bytes come from `base + lane_off` (wave-uniform 64-bit base in
// scalar registers, per-lane 32-bit byte offset) and land at LDS byte address `lds + 16 l` (`lds + 4 l`), `lds` wave-uniform
// (M0).  Checked on the device by scripts/micro/glds_test.hip: M0 addresses beyond 64 KiB work, lanes outside EXEC write
// nothing, an instruction offset would advance the global AND the LDS address (not used here), and the issuing wavefront's
// `s_waitcnt vmcnt(N)` -- counted in order together with its stores -- is what orders its own ds_read behind the copy.
// hipcc does not know these instructions exist: no wait is inserted for them (the point: a counted vmcnt(N) instead of the
// vmcnt(0) the builtin draws before the next LDS read), so EVERY read of the copied bytes must follow a spec_wait_vm.
// NL < 64: only lanes 0..NL-1 copy (EXEC is narrowed inside the statement and restored; the callers run with all lanes on).
// NT: non-temporal (streamed rows that nobody reads again should not displace the gathered node rows in the L2)
template <int NL, bool NT = false>
__device__ __forceinline__ void spec_glds16(unsigned lds, const void* base, unsigned lane_off) {
  if constexpr (NL >= 64) {
    if constexpr (NT)
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt" ::"s"(lds), "v"(lane_off), "s"(base) : "memory");
    else
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(lane_off), "s"(base) : "memory");
  } else {
    unsigned long long keep;
    if constexpr (NT)
      asm volatile("s_mov_b64 %0, exec\n\ts_mov_b32 m0, %1\n\ts_bfm_b64 exec, %4, 0\n\tglobal_load_lds_dwordx4 %2, %3 nt\n\ts_mov_b64 exec, %0"
                   : "=&s"(keep) : "s"(lds), "v"(lane_off), "s"(base), "n"(NL) : "memory");
    else
      asm volatile("s_mov_b64 %0, exec\n\ts_mov_b32 m0, %1\n\ts_bfm_b64 exec, %4, 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, %0"
                   : "=&s"(keep) : "s"(lds), "v"(lane_off), "s"(base), "n"(NL) : "memory");
  }
}
template <int NL>
__device__ __forceinline__ void spec_glds4(unsigned lds, const void* base, unsigned lane_off) {
  static_assert(NL < 64, "partial rows only");
  unsigned long long keep;
  asm volatile("s_mov_b64 %0, exec\n\ts_mov_b32 m0, %1\n\ts_bfm_b64 exec, %4, 0\n\tglobal_load_lds_dword %2, %3\n\ts_mov_b64 exec, %0"
               : "=&s"(keep) : "s"(lds), "v"(lane_off), "s"(base), "n"(NL) : "memory");
}
// all but the N most recently issued vector-memory operations of this wavefront (copies AND stores, in order) have completed
template <int N>
__device__ __forceinline__ void spec_wait_vm() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// every LDS read (and scalar load) of this wavefront has returned: a slot may be refilled
__device__ __forceinline__ void spec_wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// ---- wave64 reductions -------------------------------------------------------------------------------------
// Sum over the 64 lanes of a wavefront with DPP row operations (no LDS traffic):
//   quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_ror:4, row_ror:8  -> every lane holds its 16-lane row sum
//   then the four row sums are combined through v_readlane.
__device__ __forceinline__ float spec_row_sum(float v) {
  int t;
  t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true);
  v += __builtin_bit_cast(float, t);
  t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true);
  v += __builtin_bit_cast(float, t);
  t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true);
  v += __builtin_bit_cast(float, t);
  t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true);
  v += __builtin_bit_cast(float, t);
  return v;
}

__device__ __forceinline__ float spec_wave_sum(float v) {
  v = spec_row_sum(v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return (r0 + r1) + (r2 + r3);
}

__device__ __forceinline__ double spec_wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Reduce K per-lane values over the wavefront and store the K sums to dst[0..K).
// Generic form: one full wave reduction per value (K * ~13 VALU instructions).
template <typename T, int K>
__device__ __forceinline__ void spec_wave_reduce_store_each(const T* __restrict__ q, T* __restrict__ dst, int lane,
                                                            bool atomic = false) {
  T mine = T(0);
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const T r = spec_wave_sum(q[j]);
    if (lane == j) mine = r;
  }
  if (lane < K) {
    if (atomic) unsafeAtomicAdd(dst + lane, mine);
    else dst[lane] = mine;
  }
}

template <int CTRL>
__device__ __forceinline__ float spec_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// One halving step of the transposing reduction: NIN values per lane -> (NIN+1)/2.  Lanes whose `bit` is clear keep
// the even member of each pair and hand the odd one to their partner (lane ^ stride), and vice versa.
template <int NIN, int STEP>
__device__ __forceinline__ void spec_halve(const float* in, float* out, bool bit) {
#pragma unroll
  for (int m = 0; m < (NIN + 1) / 2; ++m) {
    const float lo = in[2 * m];
    const float hi = (2 * m + 1 < NIN) ? in[2 * m + 1] : 0.f;
    const float keep = bit ? hi : lo;
    const float send = bit ? lo : hi;
    float got;
    if (STEP == 0) got = spec_dpp<0xB1>(send);                       // quad_perm [1,0,3,2]: lane ^ 1
    else if (STEP == 1) got = spec_dpp<0x4E>(send);                  // quad_perm [2,3,0,1]: lane ^ 2
    else if (STEP == 2) got = spec_dpp<0x1B>(spec_dpp<0x141>(send));  // row_half_mirror . quad reverse: lane ^ 4
    else got = spec_dpp<0x128>(send);                                // row_ror:8: lane ^ 8
    out[m] = keep + got;
  }
}

// float, K <= 16: transposing butterfly.  Four halving steps inside each 16-lane row leave lane r of every row with
// the row-partial sum of value r; two gfx950 half-exchanges (v_permlane16_swap / v_permlane32_swap) then all-reduce
// the four rows.  ~(3K + 10) VALU instructions instead of ~13K, and a single coalesced store.
// A lane beyond the last channel of a partial chunk duplicates the clamped channel's work (all-lanes kernels): its partial
// sums must not enter a wave reduction.
template <typename T, int K>
__device__ __forceinline__ void spec_mask_dup(T* __restrict__ q, bool own) {
#pragma unroll
  for (int k = 0; k < K; ++k) q[k] = own ? q[k] : T(0);
}

// atomic: add the sums to dst instead of storing them (several wavefronts contribute to one row)
template <typename T, int K>
__device__ __forceinline__ void spec_wave_reduce_store(const T* __restrict__ q, T* __restrict__ dst, int lane, bool atomic = false) {
  if constexpr (sizeof(T) != 4 || (K > 16)) {
    spec_wave_reduce_store_each<T, K>(q, dst, lane, atomic);
  } else {
    constexpr int KA = (K + 1) / 2, KB = (KA + 1) / 2, KC = (KB + 1) / 2;
    float a[KA], b[KB], c[KC], d[1];
    spec_halve<K, 0>(q, a, (lane & 1) != 0);
    spec_halve<KA, 1>(a, b, (lane & 2) != 0);
    spec_halve<KB, 2>(b, c, (lane & 4) != 0);
    spec_halve<KC, 3>(c, d, (lane & 8) != 0);
    // all-reduce over the four rows.  v_permlane16_swap: odd rows of vdst <-> even rows of src; v_permlane32_swap: upper
    // half of vdst <-> lower half of src; with vdst = src = r the two results hold complementary rows in every lane.
    // (inline asm: two wait states are required between a VALU write of an operand and the swap.)
    float r = d[0], r2 = d[0];
    asm("v_nop\n\tv_nop\n\tv_permlane16_swap_b32 %0, %1" : "+v"(r), "+v"(r2));
    r += r2;
    r2 = r;
    asm("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(r), "+v"(r2));
    r += r2;
    if (lane < K) {
      if (atomic) unsafeAtomicAdd(dst + lane, r);
      else dst[lane] = r;
    }
  }
}

}  // namespace nqa
