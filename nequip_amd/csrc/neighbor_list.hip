// Cell-list neighbour list on the device (the caller-side step immediately before the hot path).
//
// Replaces _compute_neighborlist_single_frame (nequip/data/_nl.py:63-165), which hands the positions to a CPU library
// (matscipy `neighbour_list("ijS", ...)` by default) and copies the result back: all ordered pairs (i, j, S) with
//     | pos[j] - pos[i] + S @ cell | < r_max,      excluding i == j with S == 0,
// `edge_index[0] = i` (the convolution centre), `edge_index[1] = j`, S integer lattice shifts (0 along non-periodic
// directions), positions anywhere (not necessarily inside the cell), any triclinic cell, cells smaller than the cutoff
// (several images of the same atom), mixed periodicity.
//
// Device algorithm (float64 throughout, like the reference libraries):
//   1. plan   : fractional coordinates s = pos @ cell^-1; periodic directions are wrapped into [0,1) (the integer part is
//               remembered per atom), non-periodic ones get a bounding box.  Bins per direction = floor(extent / r_max)
//               (>= 1, coarsened until the grid fits the workspace), so a bin is at least r_max thick unless the whole
//               cell is thinner -- then ceil(r_max / thickness) bins/images are searched on either side.
//   2. bin    : bin id per atom and a histogram of the bins (the atomic's return value = the atom's arrival slot in its bin);
//               prefix sum over the bins; atoms dropped into their bins by arrival slot; one wavefront per bin then puts its
//               atoms in ascending order (rank sort: a bin holds ~10 atoms) and gathers their coordinates.  The arrival
//               order is arbitrary, the result is not: bins hold their atoms in ascending index order.
//   3. count  : one wavefront per atom walks the (2R+1)^3 neighbouring bins (with image bookkeeping), 64 candidates at a
//               time (ballot + population count), counts hits;
//               exclusive scan -> rowptr.  The host reads rowptr[N] (the one unavoidable synchronisation: E is data
//               dependent) and allocates the outputs.
//   4. fill   : same walk, writes (i, j, S) at rowptr[i] + k.  Edges come out grouped by centre atom in ascending order
//               (= the dst-sorted order the tensor-product kernels want) and deterministically ordered within an atom.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "nequip_amd.h"
#include "plan.h"

namespace nqa {

struct NLHeader {
  double cell[9];   // rows = lattice vectors (identity for a missing cell)
  double inv[9];    // cell^-1 (s = pos @ inv)
  double lo[3];     // fractional origin of the grid per direction (0 for periodic ones)
  double width[3];  // fractional width of one bin
  int32_t nb[3];    // bins per direction
  int32_t reach[3]; // bins searched on either side
  int32_t pbc[3];
  int32_t nbins;    // nb[0]*nb[1]*nb[2]
  double rmax2;
  int32_t pad_axis;  // capacity-padded lists: padding edges are self images (i <- i) shifted by +-(pad_k0 + t) cells along
  int32_t pad_k0;    // this lattice vector, pad_k0 * |a| > r_max: longer than the cutoff, so they carry no interaction
};

// workspace layout (all offsets 256-byte aligned), N atoms, B = N + 8 bins capacity
struct NLLayout {
  int64_t header, sfrac, ioff, key, val, bin_count, rowptr_bin, atom_sorted, dummy_other, s_sorted, o_sorted, counts, total;
};

static int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

static NLLayout nl_layout(int64_t N) {
  const int64_t B = N + 8;
  NLLayout L{};
  int64_t off = 0;
  auto take = [&](int64_t bytes) {
    const int64_t o = off;
    off = align256(off + bytes);
    return o;
  };
  L.header = take(sizeof(NLHeader));
  L.sfrac = take(N * 3 * 8);        // wrapped fractional coordinates
  L.ioff = take(N * 3 * 4);         // integer parts removed by the wrapping
  L.key = take(N * 4);              // bin id per atom
  L.val = take(N * 4);              // arrival slot of the atom in its bin
  L.bin_count = take((B + 1) * 4);  // atoms per bin (zeroed by the plan kernel)
  L.rowptr_bin = take((B + 1) * 4);
  L.atom_sorted = take(N * 4);      // atom indices in bin order, ascending within a bin
  L.dummy_other = take(N * 4);      // atom indices in bin order, arrival order within a bin
  L.s_sorted = take(N * 3 * 8);
  L.o_sorted = take(N * 3 * 4);
  L.counts = take((N + 1) * 4);
  L.total = off;
  return L;
}

__device__ __forceinline__ void inv3(const double* c, double* inv) {
  const double det = c[0] * (c[4] * c[8] - c[5] * c[7]) - c[1] * (c[3] * c[8] - c[5] * c[6]) +
                     c[2] * (c[3] * c[7] - c[4] * c[6]);
  const double r = 1.0 / det;
  inv[0] = (c[4] * c[8] - c[5] * c[7]) * r;
  inv[1] = (c[2] * c[7] - c[1] * c[8]) * r;
  inv[2] = (c[1] * c[5] - c[2] * c[4]) * r;
  inv[3] = (c[5] * c[6] - c[3] * c[8]) * r;
  inv[4] = (c[0] * c[8] - c[2] * c[6]) * r;
  inv[5] = (c[2] * c[3] - c[0] * c[5]) * r;
  inv[6] = (c[3] * c[7] - c[4] * c[6]) * r;
  inv[7] = (c[1] * c[6] - c[0] * c[7]) * r;
  inv[8] = (c[0] * c[4] - c[1] * c[3]) * r;
}

// One workgroup: bounding box of the fractional coordinates, then thread 0 sizes the grid.
__global__ __launch_bounds__(1024) void nl_plan_kernel(const double* __restrict__ pos, const double* __restrict__ cell,
                                                       const int32_t* __restrict__ pbc, double r_max, int64_t N,
                                                       int64_t bin_capacity, NLHeader* __restrict__ h,
                                                       int32_t* __restrict__ bin_count) {
  __shared__ double smin[3][1024], smax[3][1024];
  for (int64_t b = threadIdx.x; b <= bin_capacity; b += 1024) bin_count[b] = 0;
  __shared__ double c[9], inv[9];
  const int tid = threadIdx.x;
  if (tid == 0) {
    bool any = false;
    for (int i = 0; i < 9; ++i) {
      c[i] = cell ? cell[i] : 0.0;
      any = any || c[i] != 0.0;
    }
    // ase.geometry.complete_cell analogue for the cases the reference meets here: a missing / all-zero cell becomes the
    // identity (only legal without periodicity)
    if (!any) {
      for (int i = 0; i < 9; ++i) c[i] = (i % 4 == 0) ? 1.0 : 0.0;
    }
    inv3(c, inv);
  }
  __syncthreads();
  double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
  for (int64_t i = tid; i < N; i += 1024) {
    const double x = pos[3 * i], y = pos[3 * i + 1], z = pos[3 * i + 2];
    for (int d = 0; d < 3; ++d) {
      const double s = x * inv[d] + y * inv[3 + d] + z * inv[6 + d];
      mn[d] = fmin(mn[d], s);
      mx[d] = fmax(mx[d], s);
    }
  }
  for (int d = 0; d < 3; ++d) {
    smin[d][tid] = mn[d];
    smax[d][tid] = mx[d];
  }
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if (tid < off) {
      for (int d = 0; d < 3; ++d) {
        smin[d][tid] = fmin(smin[d][tid], smin[d][tid + off]);
        smax[d][tid] = fmax(smax[d][tid], smax[d][tid + off]);
      }
    }
    __syncthreads();
  }
  if (tid != 0) return;
  for (int i = 0; i < 9; ++i) {
    h->cell[i] = c[i];
    h->inv[i] = inv[i];
  }
  // perpendicular height of the cell along direction d = 1 / | column d of cell^-1 |
  int64_t nb[3];
  for (int d = 0; d < 3; ++d) {
    const double height = 1.0 / sqrt(inv[d] * inv[d] + inv[3 + d] * inv[3 + d] + inv[6 + d] * inv[6 + d]);
    const bool per = pbc != nullptr && pbc[d] != 0;
    h->pbc[d] = per ? 1 : 0;
    double lo, extent;  // fractional
    if (per) {
      lo = 0.0;
      extent = 1.0;
    } else {
      lo = N > 0 ? smin[d][0] : 0.0;
      extent = N > 0 ? (smax[d][0] - smin[d][0]) : 0.0;
      extent = extent * (1.0 + 1e-9) + 1e-9;  // the topmost atom must fall inside the last bin
    }
    int64_t n = (int64_t)floor(extent * height / r_max);
    if (n < 1) n = 1;
    if (n > 1024) n = 1024;
    nb[d] = n;
    h->lo[d] = lo;
    h->width[d] = extent;  // divided by nb below
    h->reach[d] = 0;       // filled below (needs the final nb)
    h->rmax2 = height;     // scratch; overwritten below
    smin[d][1] = height;   // keep the height for the second pass
  }
  while (nb[0] * nb[1] * nb[2] > bin_capacity) {  // coarser bins are always valid
    int big = 0;
    if (nb[1] > nb[big]) big = 1;
    if (nb[2] > nb[big]) big = 2;
    nb[big] = (nb[big] + 1) / 2;
  }
  for (int d = 0; d < 3; ++d) {
    const double height = smin[d][1];
    const double extent = h->width[d];
    h->nb[d] = (int32_t)nb[d];
    h->width[d] = extent / (double)nb[d];
    const double thick = h->width[d] * height;  // real-space thickness of one bin
    int reach = (int)ceil(r_max / thick);
    if (!h->pbc[d] && reach > nb[d] - 1) reach = (int)(nb[d] - 1);  // nothing beyond the box
    h->reach[d] = reach;
  }
  h->nbins = (int32_t)(nb[0] * nb[1] * nb[2]);
  h->rmax2 = r_max * r_max;
  // padding edges (nqa_neighbor_list_fill_padded): the shortest lattice vector among the periodic directions (any direction
  // when there is none), repeated often enough to leave the cutoff sphere
  int axis = -1;
  double best = 0.0;
  for (int pass = 0; pass < 2 && axis < 0; ++pass) {
    for (int d = 0; d < 3; ++d) {
      if (pass == 0 && !h->pbc[d]) continue;
      const double len = sqrt(c[3 * d] * c[3 * d] + c[3 * d + 1] * c[3 * d + 1] + c[3 * d + 2] * c[3 * d + 2]);
      if (axis < 0 || len < best) {
        axis = d;
        best = len;
      }
    }
  }
  h->pad_axis = axis;
  h->pad_k0 = (int32_t)floor(r_max / best) + 1;
}

__global__ __launch_bounds__(256) void nl_bin_kernel(const double* __restrict__ pos, int64_t N,
                                                     const NLHeader* __restrict__ h, double* __restrict__ sfrac,
                                                     int32_t* __restrict__ ioff, int32_t* __restrict__ key,
                                                     int32_t* __restrict__ val, int32_t* __restrict__ bin_count) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const double x = pos[3 * i], y = pos[3 * i + 1], z = pos[3 * i + 2];
  int b[3];
  for (int d = 0; d < 3; ++d) {
    double s = x * h->inv[d] + y * h->inv[3 + d] + z * h->inv[6 + d];
    int o = 0;
    if (h->pbc[d]) {
      const double fl = floor(s);
      o = (int)fl;
      s -= fl;
      if (s >= 1.0) {  // s was -epsilon: rounds to 1.0
        s -= 1.0;
        o += 1;
      }
    }
    int bi = (int)floor((s - h->lo[d]) / h->width[d]);
    if (bi < 0) bi = 0;
    if (bi > h->nb[d] - 1) bi = h->nb[d] - 1;
    b[d] = bi;
    sfrac[3 * i + d] = s;
    ioff[3 * i + d] = o;
  }
  const int32_t bin = (b[0] * h->nb[1] + b[1]) * h->nb[2] + b[2];
  key[i] = bin;
  val[i] = atomicAdd(&bin_count[bin], 1);
}

__global__ __launch_bounds__(256) void nl_place_kernel(int64_t N, const int32_t* __restrict__ key,
                                                       const int32_t* __restrict__ val,
                                                       const int32_t* __restrict__ rowptr_bin,
                                                       int32_t* __restrict__ atom_arrival) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < N) atom_arrival[rowptr_bin[key[i]] + val[i]] = (int32_t)i;
}

// One wavefront per bin: rank of every atom of the bin among the bin's atom indices = its final position; coordinates follow.
__global__ __launch_bounds__(256) void nl_order_kernel(int64_t B, const int32_t* __restrict__ rowptr_bin,
                                                       const int32_t* __restrict__ atom_arrival,
                                                       const double* __restrict__ sfrac, const int32_t* __restrict__ ioff,
                                                       int32_t* __restrict__ atom_sorted, double* __restrict__ s_sorted,
                                                       int32_t* __restrict__ o_sorted) {
  const int64_t b = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (b >= B) return;
  const int lane = threadIdx.x & 63;
  const int k0 = rowptr_bin[b], k1 = rowptr_bin[b + 1];
  for (int kb = k0; kb < k1; kb += 64) {
    const int k = kb + lane;
    const int a = k < k1 ? atom_arrival[k] : 0x7fffffff;
    int rank = 0;
    for (int t = k0; t < k1; ++t) rank += atom_arrival[t] < a ? 1 : 0;
    if (k < k1) {
      const int pos = k0 + rank;
      atom_sorted[pos] = a;
      for (int d = 0; d < 3; ++d) {
        s_sorted[3 * (int64_t)pos + d] = sfrac[3 * (int64_t)a + d];
        o_sorted[3 * (int64_t)pos + d] = ioff[3 * (int64_t)a + d];
      }
    }
  }
}

// Shared walk of count and fill, one WAVEFRONT per atom: the neighbouring bins are visited in a fixed order and the atoms of a
// bin are tested 64 at a time, one per lane; a hit's position in the atom's edge row is the number of hits before it (ballot +
// population count), i.e. exactly the order in which a single thread walking the same bins would emit them.
// FILL == false: returns the number of neighbours of atom i (the same value in every lane).
template <bool FILL>
__device__ __forceinline__ int nl_walk(int64_t i, int lane, const NLHeader* __restrict__ h, const double* __restrict__ sfrac,
                                       const int32_t* __restrict__ ioff, const int32_t* __restrict__ rowptr_bin,
                                       const int32_t* __restrict__ atom_sorted, const double* __restrict__ s_sorted,
                                       const int32_t* __restrict__ o_sorted, int64_t base, int64_t E,
                                       int64_t* __restrict__ edge_index, double* __restrict__ shift,
                                       int32_t* __restrict__ src32 = nullptr) {
  const double si[3] = {sfrac[3 * i], sfrac[3 * i + 1], sfrac[3 * i + 2]};
  const int oi[3] = {ioff[3 * i], ioff[3 * i + 1], ioff[3 * i + 2]};
  int bi[3];
  for (int d = 0; d < 3; ++d) {
    int b = (int)floor((si[d] - h->lo[d]) / h->width[d]);
    b = b < 0 ? 0 : (b > h->nb[d] - 1 ? h->nb[d] - 1 : b);
    bi[d] = b;
  }
  const double* __restrict__ c = h->cell;
  int cnt = 0;
  for (int dx = -h->reach[0]; dx <= h->reach[0]; ++dx) {
    int bx = bi[0] + dx, nx = 0;
    if (h->pbc[0]) {
      nx = (bx >= 0) ? bx / h->nb[0] : -((-bx + h->nb[0] - 1) / h->nb[0]);
      bx -= nx * h->nb[0];
    } else if (bx < 0 || bx >= h->nb[0]) {
      continue;
    }
    for (int dy = -h->reach[1]; dy <= h->reach[1]; ++dy) {
      int by = bi[1] + dy, ny = 0;
      if (h->pbc[1]) {
        ny = (by >= 0) ? by / h->nb[1] : -((-by + h->nb[1] - 1) / h->nb[1]);
        by -= ny * h->nb[1];
      } else if (by < 0 || by >= h->nb[1]) {
        continue;
      }
      // bins that follow each other along z lie next to each other in the sorted atom array: the z range is walked as runs
      // of bins of one periodic image (one run, two across a cell boundary, more only for cells thinner than the cutoff), 64
      // candidates of a run at a time -- the same candidate order as bin by bin, with fuller wavefronts
      int z = bi[2] - h->reach[2], zhi = bi[2] + h->reach[2];
      if (!h->pbc[2]) {
        z = z < 0 ? 0 : z;
        zhi = zhi > h->nb[2] - 1 ? h->nb[2] - 1 : zhi;
      }
      while (z <= zhi) {
        int nz = 0;
        if (h->pbc[2]) nz = (z >= 0) ? z / h->nb[2] : -((-z + h->nb[2] - 1) / h->nb[2]);
        int zend = (nz + 1) * h->nb[2] - 1;  // last bin of this image
        zend = zend < zhi ? zend : zhi;
        const int64_t row = ((int64_t)bx * h->nb[1] + by) * h->nb[2] - (int64_t)nz * h->nb[2];
        const int k1 = rowptr_bin[row + zend + 1];
        for (int kb = rowptr_bin[row + z]; kb < k1; kb += 64) {
          const int k = kb + lane;
          bool hit = false;
          int j = 0;
          if (k < k1) {
            // (s_j - s_i) + n: the reverse edge computes (s_i - s_j) - n, the exact negative in floating point, so both
            // directions of a pair see the SAME squared length and the list is symmetric by construction
            const double fx = (s_sorted[3 * k] - si[0]) + nx;
            const double fy = (s_sorted[3 * k + 1] - si[1]) + ny;
            const double fz = (s_sorted[3 * k + 2] - si[2]) + nz;
            const double rx = fx * c[0] + fy * c[3] + fz * c[6];
            const double ry = fx * c[1] + fy * c[4] + fz * c[7];
            const double rz = fx * c[2] + fy * c[5] + fz * c[8];
            const double r2 = rx * rx + ry * ry + rz * rz;
            j = atom_sorted[k];
            hit = (r2 < h->rmax2) && !(j == i && nx == 0 && ny == 0 && nz == 0);
          }
          const uint64_t m = __builtin_amdgcn_ballot_w64(hit);
          if (FILL && hit) {
            const int64_t e = base + cnt + __popcll(m & ((1ull << lane) - 1ull));
            edge_index[e] = i;
            edge_index[E + e] = j;
            if (src32 != nullptr) src32[e] = j;
            // pos_j - pos_i + S @ cell = (s_j + n - s_i) @ cell with pos = (s + o) @ cell  =>  S = n - o_j + o_i
            shift[3 * e + 0] = (double)(nx - o_sorted[3 * k] + oi[0]);
            shift[3 * e + 1] = (double)(ny - o_sorted[3 * k + 1] + oi[1]);
            shift[3 * e + 2] = (double)(nz - o_sorted[3 * k + 2] + oi[2]);
          }
          cnt += __popcll(m);
        }
        z = zend + 1;
      }
    }
  }
  return cnt;
}

__global__ __launch_bounds__(256) void nl_count_kernel(int64_t N, const NLHeader* __restrict__ h,
                                                       const double* __restrict__ sfrac, const int32_t* __restrict__ ioff,
                                                       const int32_t* __restrict__ rowptr_bin,
                                                       const int32_t* __restrict__ atom_sorted,
                                                       const double* __restrict__ s_sorted,
                                                       const int32_t* __restrict__ o_sorted, int32_t* __restrict__ counts) {
  const int64_t i = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // one wavefront per atom
  if (i >= N) return;
  const int lane = threadIdx.x & 63;
  const int cnt = nl_walk<false>(i, lane, h, sfrac, ioff, rowptr_bin, atom_sorted, s_sorted, o_sorted, 0, 0, nullptr, nullptr);
  if (lane == 0) counts[i] = cnt;
}

// exclusive scan of counts[0..N) into rowptr[0..N], one workgroup: every thread sums a contiguous slice, the 1024 slice sums
// are scanned across the workgroup (wavefront shuffles + one pass over the 16 wavefront totals), every thread rewrites its slice
__global__ __launch_bounds__(1024) void nl_scan_kernel(int64_t N, const int32_t* __restrict__ counts,
                                                       int32_t* __restrict__ rowptr, int32_t* __restrict__ overflow) {
  __shared__ int64_t wave_total[16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t per = (N + 1023) / 1024;
  const int64_t lo = tid * per < N ? tid * per : N, hi = lo + per < N ? lo + per : N;
  int64_t sum = 0;
  for (int64_t i = lo; i < hi; ++i) sum += counts[i];
  int64_t incl = sum;  // inclusive scan over the wavefront
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int64_t t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  if (lane == 63) wave_total[wv] = incl;
  __syncthreads();
  int64_t before = 0;
  for (int w = 0; w < wv; ++w) before += wave_total[w];
  int64_t run = before + incl - sum;
  for (int64_t i = lo; i < hi; ++i) {
    rowptr[i] = (int32_t)run;
    run += counts[i];
  }
  if (tid == 1023) {
    const int64_t total = before + incl;
    rowptr[N] = (int32_t)total;
    if (total > 2147483647LL && overflow != nullptr) *overflow = 1;
  }
}

__global__ __launch_bounds__(256) void nl_fill_kernel(int64_t N, int64_t E, const NLHeader* __restrict__ h,
                                                      const double* __restrict__ sfrac, const int32_t* __restrict__ ioff,
                                                      const int32_t* __restrict__ rowptr_bin,
                                                      const int32_t* __restrict__ atom_sorted,
                                                      const double* __restrict__ s_sorted,
                                                      const int32_t* __restrict__ o_sorted,
                                                      const int32_t* __restrict__ rowptr, int64_t* __restrict__ edge_index,
                                                      double* __restrict__ shift) {
  const int64_t i = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // one wavefront per atom
  if (i >= N) return;
  nl_walk<true>(i, (int)(threadIdx.x & 63), h, sfrac, ioff, rowptr_bin, atom_sorted, s_sorted, o_sorted, rowptr[i], E,
                edge_index, shift);
}

// ---- capacity-padded list (no host read-back of the edge count: the whole MD step can live in one hipGraph) -----------------
// E_cap - E_real padding edges (E_cap even) are dealt out to the atoms as self-image PAIRS (i <- i, +S_t), (i <- i, -S_t),
// S_t = (pad_k0 + t) cells along pad_axis: atom i gets q + (i < rem) pairs, q = pairs / N, rem = pairs % N, behind its real
// edges.  They are longer than r_max, i.e. outside the polynomial cutoff: zero radial embedding, zero weights (the radial MLP is
// bias-free), zero derivative.  A list that does not fit (E_real > E_cap, or E_cap - E_real odd) is replaced by padding only
// and reported through status[0]: every index stays in range, the caller re-runs with a larger capacity.
struct NLPadPlan {
  bool bad;
  int64_t q, rem;
};
__device__ __forceinline__ NLPadPlan nl_pad_plan(const int32_t* __restrict__ rowptr, int64_t N, int64_t E_cap) {
  const int64_t E_real = rowptr[N];
  int64_t tail = E_cap - E_real;
  NLPadPlan p;
  p.bad = tail < 0 || (tail & 1) != 0;
  if (p.bad) tail = E_cap;
  const int64_t pairs = tail / 2;
  p.q = pairs / N;
  p.rem = pairs % N;
  return p;
}
__device__ __forceinline__ int64_t nl_pads_before(const NLPadPlan& p, int64_t i) { return 2 * (i * p.q + (i < p.rem ? i : p.rem)); }

__global__ __launch_bounds__(256) void nl_pad_rowptr_kernel(int64_t N, int64_t E_cap, const int32_t* __restrict__ rowptr,
                                                            int32_t* __restrict__ rowptr_out, int32_t* __restrict__ status) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i > N) return;
  const NLPadPlan p = nl_pad_plan(rowptr, N, E_cap);
  rowptr_out[i] = (int32_t)((p.bad ? 0 : (int64_t)rowptr[i]) + nl_pads_before(p, i));
  if (i == 0 && status != nullptr) {
    status[0] = p.bad ? 1 : 0;
    status[1] = rowptr[N];
  }
}

__global__ __launch_bounds__(256) void nl_fill_padded_kernel(int64_t N, int64_t E_cap, const NLHeader* __restrict__ h,
                                                             const double* __restrict__ sfrac, const int32_t* __restrict__ ioff,
                                                             const int32_t* __restrict__ rowptr_bin,
                                                             const int32_t* __restrict__ atom_sorted,
                                                             const double* __restrict__ s_sorted,
                                                             const int32_t* __restrict__ o_sorted,
                                                             const int32_t* __restrict__ rowptr,
                                                             int64_t* __restrict__ edge_index, double* __restrict__ shift,
                                                             int32_t* __restrict__ src32) {
  const int64_t i = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // one wavefront per atom
  if (i >= N) return;
  const int lane = threadIdx.x & 63;
  const NLPadPlan p = nl_pad_plan(rowptr, N, E_cap);
  const int64_t base = (p.bad ? 0 : (int64_t)rowptr[i]) + nl_pads_before(p, i);
  int cnt = 0;
  if (!p.bad)
    cnt = nl_walk<true>(i, lane, h, sfrac, ioff, rowptr_bin, atom_sorted, s_sorted, o_sorted, base, E_cap, edge_index, shift,
                        src32);
  const int64_t npad = p.q + (i < p.rem ? 1 : 0);
  const int axis = h->pad_axis;
  for (int64_t t = lane; t < npad; t += 64) {
    const double k = (double)(h->pad_k0 + t);
    for (int sgn = 0; sgn < 2; ++sgn) {
      const int64_t e = base + cnt + 2 * t + sgn;
      edge_index[e] = i;
      edge_index[E_cap + e] = i;
      if (src32 != nullptr) src32[e] = (int32_t)i;
      for (int d = 0; d < 3; ++d) shift[3 * e + d] = d == axis ? (sgn ? -k : k) : 0.0;
    }
  }
}

static int nl_status(const char* fn) {
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_error(std::string(fn) + ": " + hipGetErrorString(err));
    return NQA_ERR_LAUNCH;
  }
  return NQA_OK;
}

}  // namespace nqa

using namespace nqa;

extern "C" {

int64_t nqa_neighbor_list_workspace_bytes(int64_t num_atoms) {
  if (num_atoms < 0) return -1;
  return nl_layout(num_atoms).total;
}

int nqa_neighbor_list_count(const double* pos, const double* cell, const int32_t* pbc, double r_max, int64_t num_atoms,
                            void* workspace, int64_t workspace_bytes, int32_t* rowptr, nqa_stream stream) {
  if (num_atoms < 0 || !(r_max > 0.0) || !rowptr || (num_atoms > 0 && !pos)) {
    set_error("nqa_neighbor_list_count: invalid argument");
    return NQA_ERR_INVALID;
  }
  const NLLayout L = nl_layout(num_atoms);
  if (!workspace || workspace_bytes < L.total) {
    set_error("nqa_neighbor_list_count: workspace missing or too small");
    return NQA_ERR_WORKSPACE;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  char* w = static_cast<char*>(workspace);
  const int64_t N = num_atoms, B = N + 8;
  NLHeader* h = reinterpret_cast<NLHeader*>(w + L.header);
  double* sfrac = reinterpret_cast<double*>(w + L.sfrac);
  int32_t* ioff = reinterpret_cast<int32_t*>(w + L.ioff);
  int32_t* key = reinterpret_cast<int32_t*>(w + L.key);
  int32_t* val = reinterpret_cast<int32_t*>(w + L.val);
  int32_t* bin_count = reinterpret_cast<int32_t*>(w + L.bin_count);
  int32_t* rowptr_bin = reinterpret_cast<int32_t*>(w + L.rowptr_bin);
  int32_t* atom_sorted = reinterpret_cast<int32_t*>(w + L.atom_sorted);
  int32_t* atom_arrival = reinterpret_cast<int32_t*>(w + L.dummy_other);
  double* s_sorted = reinterpret_cast<double*>(w + L.s_sorted);
  int32_t* o_sorted = reinterpret_cast<int32_t*>(w + L.o_sorted);
  int32_t* counts = reinterpret_cast<int32_t*>(w + L.counts);
  hipLaunchKernelGGL(nl_plan_kernel, dim3(1), dim3(1024), 0, s, pos, cell, pbc, r_max, N, B, h, bin_count);
  if (N > 0) {
    const unsigned g256 = (unsigned)((N + 255) / 256);
    hipLaunchKernelGGL(nl_bin_kernel, dim3(g256), dim3(256), 0, s, pos, N, h, sfrac, ioff, key, val, bin_count);
    hipLaunchKernelGGL(nl_scan_kernel, dim3(1), dim3(1024), 0, s, B, bin_count, rowptr_bin, (int32_t*)nullptr);
    hipLaunchKernelGGL(nl_place_kernel, dim3(g256), dim3(256), 0, s, N, key, val, rowptr_bin, atom_arrival);
    hipLaunchKernelGGL(nl_order_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, s, B, rowptr_bin, atom_arrival, sfrac,
                       ioff, atom_sorted, s_sorted, o_sorted);
    hipLaunchKernelGGL(nl_count_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, N, h, sfrac, ioff,
                       rowptr_bin, atom_sorted, s_sorted, o_sorted, counts);
  }
  hipLaunchKernelGGL(nl_scan_kernel, dim3(1), dim3(1024), 0, s, N, counts, rowptr, (int32_t*)nullptr);
  return nl_status("nqa_neighbor_list_count");
}

int nqa_neighbor_list_fill(const void* workspace, const int32_t* rowptr, int64_t num_atoms, int64_t num_edges,
                           int64_t* edge_index, double* edge_cell_shift, nqa_stream stream) {
  if (num_atoms < 0 || num_edges < 0 || !workspace || !rowptr ||
      (num_edges > 0 && (!edge_index || !edge_cell_shift))) {
    set_error("nqa_neighbor_list_fill: invalid argument");
    return NQA_ERR_INVALID;
  }
  if (num_atoms == 0 || num_edges == 0) return NQA_OK;
  const NLLayout L = nl_layout(num_atoms);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const char* w = static_cast<const char*>(workspace);
  hipLaunchKernelGGL(nl_fill_kernel, dim3((unsigned)((num_atoms + 3) / 4)), dim3(256), 0, s, num_atoms, num_edges,
                     reinterpret_cast<const NLHeader*>(w + L.header), reinterpret_cast<const double*>(w + L.sfrac),
                     reinterpret_cast<const int32_t*>(w + L.ioff), reinterpret_cast<const int32_t*>(w + L.rowptr_bin),
                     reinterpret_cast<const int32_t*>(w + L.atom_sorted), reinterpret_cast<const double*>(w + L.s_sorted),
                     reinterpret_cast<const int32_t*>(w + L.o_sorted), rowptr, edge_index, edge_cell_shift);
  return nl_status("nqa_neighbor_list_fill");
}

int nqa_neighbor_list_fill_padded(const void* workspace, const int32_t* rowptr, int64_t num_atoms, int64_t edge_capacity,
                                  int32_t* rowptr_padded, int64_t* edge_index, double* edge_cell_shift, int32_t* src_sorted,
                                  int32_t* status, nqa_stream stream) {
  if (num_atoms <= 0 || edge_capacity < 0 || (edge_capacity & 1) != 0 || edge_capacity > 2147483646LL || !workspace ||
      !rowptr || !rowptr_padded || (edge_capacity > 0 && (!edge_index || !edge_cell_shift))) {
    set_error("nqa_neighbor_list_fill_padded: invalid argument (needs atoms and an even capacity below 2^31)");
    return NQA_ERR_INVALID;
  }
  const NLLayout L = nl_layout(num_atoms);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const char* w = static_cast<const char*>(workspace);
  hipLaunchKernelGGL(nl_pad_rowptr_kernel, dim3((unsigned)((num_atoms + 1 + 255) / 256)), dim3(256), 0, s, num_atoms,
                     edge_capacity, rowptr, rowptr_padded, status);
  hipLaunchKernelGGL(nl_fill_padded_kernel, dim3((unsigned)((num_atoms + 3) / 4)), dim3(256), 0, s, num_atoms, edge_capacity,
                     reinterpret_cast<const NLHeader*>(w + L.header), reinterpret_cast<const double*>(w + L.sfrac),
                     reinterpret_cast<const int32_t*>(w + L.ioff), reinterpret_cast<const int32_t*>(w + L.rowptr_bin),
                     reinterpret_cast<const int32_t*>(w + L.atom_sorted), reinterpret_cast<const double*>(w + L.s_sorted),
                     reinterpret_cast<const int32_t*>(w + L.o_sorted), rowptr, edge_index, edge_cell_shift, src_sorted);
  return nl_status("nqa_neighbor_list_fill_padded");
}

}  // extern "C"
