// Internal: interface between the generated structure-specialised kernels (gen_spec.py) and the dispatcher.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

namespace nqa {

template <typename T>
struct SpecArgs {
  const T* __restrict__ x;   // [N, din]        (fwd, bwd_edge)
  const T* __restrict__ y;   // [E, S]
  const T* __restrict__ w;   // [E, wn]
  const T* __restrict__ g;   // [N, dout]       (bwd_edge, bwd_x)
  T* __restrict__ out;       // fwd: [N, dout]; bwd_x: [N, din]
  T* __restrict__ gw;        // [E, wn] or null
  T* __restrict__ gy;        // [E, gy_stride] or null (gy itself when mul <= 64, else per-chunk partials)
  const int32_t* __restrict__ rowptr;
  const int32_t* __restrict__ eid;
  const int32_t* __restrict__ nbr;
  int32_t N;
  int32_t mul;
  int32_t din, dout, wn;
  int32_t gy_stride;
};

// which: 0 = fwd, 1 = bwd_edge, 2 = bwd_x;  wpn: requested wavefronts per (node, chunk)
using SpecLaunchFn = int (*)(int which, int wpn, const SpecArgs<float>& a, hipStream_t stream);

struct SpecEntry {
  std::string key;
  SpecLaunchFn launch;
  int xd, s, od, np;
  SpecEntry* next;
};

SpecEntry*& spec_registry_head();
const SpecEntry* find_spec(const std::string& key);

struct SpecRegistrar {
  SpecEntry entry;
  SpecRegistrar(const char* key, SpecLaunchFn fn, int xd, int s, int od, int np) {
    entry.key = key;
    entry.launch = fn;
    entry.xd = xd;
    entry.s = s;
    entry.od = od;
    entry.np = np;
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

// Reduce K per-lane values over the wavefront and store the K sums to dst[0..K) (one store per value by the
// lane whose index equals the value index, so the K stores coalesce into one transaction for K <= 64).
template <typename T, int K>
__device__ __forceinline__ void spec_wave_reduce_store(const T* __restrict__ q, T* __restrict__ dst, int lane) {
  T mine = T(0);
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const T r = spec_wave_sum(q[j]);
    if (lane == j) mine = r;
  }
  if (lane < K) dst[lane] = mine;
}

}  // namespace nqa
