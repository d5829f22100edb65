// Laboratory harness for the pair-centric backward (bwd_pair_kernel + gx_rows_sum_kernel) of ONE generated structure,
// outside torch:  hipcc -DSPEC_FILE='"<generated .hip>"' [-DLAB_SPLIT] ... pair_lab.hip -o pair_lab_<variant>
//
//   pair_lab <topology dump (scripts/micro/dump_topo.py)> [reps] [mul] [relabel: 0 none | 1 morton | 2 random] [wpn] [mode] [morton cell] [extra LDS KiB]
//
// further positional arguments: [pair rows: 0 as built | 1 numbered in owner-slot order (w / grad_w walked sequentially)]
//   [grad_x of the other node: 0 one row per pair + row sum | 1 atomic adds into a zeroed [N, dim_in1] accumulator (needs a
//   generator variant with NQA_GEN_PAIR_GX_ATOMIC=1, or the ring kernel) + the row sum over ONE row per node]
// mode 0: pair kernel + row sum (what nqa_tp_scatter_bwd_pairs launches); 1: pair kernel only; 2: row sum only
// Prints the average duration of each kernel (HIP events, one kernel per event pair) and checksums of the results, so that
// variants built from differently generated files can be compared for speed AND for equality of what they compute.
#define NQA_LAB 1
#include SPEC_FILE

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#define CK(x)                                                                                \
  do {                                                                                       \
    hipError_t e_ = (x);                                                                     \
    if (e_ != hipSuccess) {                                                                  \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));      \
      exit(1);                                                                               \
    }                                                                                        \
  } while (0)

using namespace nqa;

template <typename T>
static T* to_dev(const std::vector<T>& v) {
  T* d = nullptr;
  CK(hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(T)));
  if (!v.empty()) CK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

static std::vector<float> randn(size_t n, unsigned seed) {
  std::mt19937 rng(seed);
  std::normal_distribution<float> d(0.f, 1.f);
  std::vector<float> v(n);
  for (auto& a : v) a = d(rng);
  return v;
}

static double checksum(const float* d, size_t n) {
  std::vector<float> h(n);
  CK(hipMemcpy(h.data(), d, n * sizeof(float), hipMemcpyDeviceToHost));
  double s = 0, s2 = 0;
  for (size_t i = 0; i < n; ++i) {
    s += h[i] * (double)((i % 7) + 1);
    s2 += (double)h[i] * h[i];
  }
  return s + 1e-3 * s2;
}

static unsigned morton3(unsigned x, unsigned y, unsigned z) {
  auto spread = [](unsigned v) {
    unsigned r = 0;
    for (int b = 0; b < 10; ++b) r |= ((v >> b) & 1u) << (3 * b);
    return r;
  };
  return spread(x) | (spread(y) << 1) | (spread(z) << 2);
}

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: pair_lab topo.bin [reps] [mul] [relabel] [wpn] [mode]\n");
    return 2;
  }
  const int reps = argc > 2 ? atoi(argv[2]) : 20;
  const int mul = argc > 3 ? atoi(argv[3]) : 64;
  const int relabel = argc > 4 ? atoi(argv[4]) : 0;
  const int wpn = argc > 5 ? atoi(argv[5]) : 4;
  const int mode = argc > 6 ? atoi(argv[6]) : 0;
  const size_t extra_lds = argc > 8 ? (size_t)atoi(argv[8]) * 1024 : 0;  // occupancy limiter: KiB of LDS per workgroup
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 1; }
  int hdr[4];
  if (fread(hdr, 4, 4, f) != 4) return 1;
  const int N = hdr[0], E = hdr[1], P = hdr[2];
  auto rd = [&](size_t n) { std::vector<int> v(n); if (fread(v.data(), 4, n, f) != n) { fprintf(stderr, "short read\n"); exit(1); } return v; };
  std::vector<int> orow = rd(N + 1), oth = rd(P), prow = rd(P), ein = rd(P), eout = rd(P), trow = rd(N + 1), tslot = rd(P);
  std::vector<double> pos(3 * (size_t)N), cell(9);
  if (fread(pos.data(), 8, pos.size(), f) != pos.size() || fread(cell.data(), 8, 9, f) != 9) { fprintf(stderr, "short read (pos)\n"); return 1; }
  fclose(f);

  const int prid = argc > 9 ? atoi(argv[9]) : 0;
  const int gxat = argc > 10 ? atoi(argv[10]) : 0;
  {
    double d = 0;
    for (int s = 0; s < P; ++s) d += std::abs((double)prow[s] - s);
    fprintf(stderr, "mean |pair_row - owner slot| = %.0f rows of %d\n", d / P, P);
  }
  const int own_rule = argc > 11 ? atoi(argv[11]) : 0;  // 0 as built (balanced parity rule) | 1 the smaller index owns | 2 the larger
  if (own_rule) {
    struct Rec { int o, j, row, ein, eout; };
    std::vector<Rec> recs(P);
    for (int n = 0; n < N; ++n)
      for (int s = orow[n]; s < orow[n + 1]; ++s) {
        Rec r{n, oth[s], prow[s], ein[s], eout[s]};
        const bool swap = own_rule == 1 ? r.j < r.o : r.j > r.o;
        if (swap) r = Rec{r.j, r.o, r.row, r.eout, r.ein};
        recs[s] = r;
      }
    std::stable_sort(recs.begin(), recs.end(), [](const Rec& a, const Rec& b) { return a.o != b.o ? a.o < b.o : a.j < b.j; });
    std::fill(orow.begin(), orow.end(), 0);
    for (int s = 0; s < P; ++s) { orow[recs[s].o + 1]++; oth[s] = recs[s].j; prow[s] = recs[s].row; ein[s] = recs[s].ein; eout[s] = recs[s].eout; }
    for (int n = 0; n < N; ++n) orow[n + 1] += orow[n];
    std::fill(trow.begin(), trow.end(), 0);
    for (int s = 0; s < P; ++s) trow[oth[s] + 1]++;
    for (int n = 0; n < N; ++n) trow[n + 1] += trow[n];
    std::vector<int> fillp(trow.begin(), trow.end() - 1);
    for (int s = 0; s < P; ++s) tslot[fillp[oth[s]]++] = s;
    int mx = 0; for (int n = 0; n < N; ++n) mx = std::max(mx, orow[n + 1] - orow[n]);
    fprintf(stderr, "ownership rule %d: most pairs of one owner %d (mean %.1f)\n", own_rule, mx, (double)P / N);
  }
  if (prid)
    for (int s = 0; s < P; ++s) prow[s] = s;
  if (relabel) {  // renumber the nodes (what the lists would be had the atoms arrived in that order)
    std::vector<int> order(N);  // order[new] = old
    std::iota(order.begin(), order.end(), 0);
    if (relabel == 1) {
      const double cs = argc > 7 ? atof(argv[7]) : 2.25;
      std::vector<unsigned> code(N);
      for (int i = 0; i < N; ++i)
        code[i] = morton3((unsigned)std::max(0.0, pos[3 * i] / cs), (unsigned)std::max(0.0, pos[3 * i + 1] / cs), (unsigned)std::max(0.0, pos[3 * i + 2] / cs));
      std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return code[a] < code[b]; });
    } else {
      std::mt19937 rng(1);
      std::shuffle(order.begin(), order.end(), rng);
    }
    std::vector<int> rank(N);
    for (int i = 0; i < N; ++i) rank[order[i]] = i;
    std::vector<int> orow2(N + 1, 0), oth2(P), prow2(P), ein2(P), eout2(P), newslot(P);
    int k = 0;
    for (int n2 = 0; n2 < N; ++n2) {
      const int n = order[n2];
      orow2[n2] = k;
      // within an owner by the (new) number of the other node
      std::vector<int> sl(orow[n + 1] - orow[n]);
      std::iota(sl.begin(), sl.end(), orow[n]);
      std::stable_sort(sl.begin(), sl.end(), [&](int a, int b) { return rank[oth[a]] < rank[oth[b]]; });
      for (int s : sl) { oth2[k] = rank[oth[s]]; prow2[k] = prow[s]; ein2[k] = ein[s]; eout2[k] = eout[s]; newslot[s] = k; ++k; }
    }
    orow2[N] = k;
    std::vector<int> trow2(N + 1, 0), tslot2(P);
    k = 0;
    for (int n2 = 0; n2 < N; ++n2) {
      const int n = order[n2];
      trow2[n2] = k;
      std::vector<int> sl;
      for (int s = trow[n]; s < trow[n + 1]; ++s) sl.push_back(newslot[tslot[s]]);
      std::sort(sl.begin(), sl.end());
      for (int s : sl) tslot2[k++] = s;
    }
    trow2[N] = k;
    orow = orow2; oth = oth2; prow = prow2; ein = ein2; eout = eout2; trow = trow2; tslot = tslot2;
  }

  const int din = kXD * mul, dout = kOD * mul, wn = kNP * mul;
#ifdef LAB_SPLIT
  const int parts = kPairParts;
#else
  const int parts = 1;
#endif
  const int nchunk = (mul + 63) / 64;
  const int gyn = nchunk * parts;
  float* x = to_dev(randn((size_t)N * din, 1));
  float* g = to_dev(randn((size_t)N * dout, 2));
  float* y = to_dev(randn((size_t)E * kS, 3));
  float* w = to_dev(randn((size_t)P * wn, 4));
  float *gw, *gy, *gxe, *out;
  CK(hipMalloc(&gw, (size_t)P * wn * 4));
  CK(hipMalloc(&gy, (size_t)E * kS * gyn * 4));
  const size_t gxe_n = gxat == 2 ? (size_t)8 * N * din : (size_t)P * din;  // gxat 2: one accumulator per XCD
  CK(hipMalloc(&gxe, gxe_n * 4));
  CK(hipMalloc(&out, (size_t)N * din * 4));
  CK(hipMemset(gw, 0, (size_t)P * wn * 4));
  CK(hipMemset(gy, 0, (size_t)E * kS * gyn * 4));
  CK(hipMemset(gxe, 0, gxe_n * 4));
  CK(hipMemset(out, 0, (size_t)N * din * 4));
  if (gxat) {  // one accumulator row per node: the "row sum" walks exactly that row
    trow.resize(N + 1); tslot.resize(N);
    std::iota(trow.begin(), trow.end(), 0); std::iota(tslot.begin(), tslot.end(), 0);
  }
  int *d_orow = to_dev(orow), *d_oth = to_dev(oth), *d_prow = to_dev(prow), *d_ein = to_dev(ein), *d_eout = to_dev(eout),
      *d_trow = to_dev(trow), *d_tslot = to_dev(tslot);

  SpecArgs<float> a{};
  a.N = N; a.mul = mul; a.din = din; a.dout = dout; a.wn = wn;
  a.x = x; a.y = y; a.w = w; a.g = g; a.gw = gw; a.gxe = gxe; a.out = out; a.gy = gy; a.gy_stride = kS * gyn;
  a.rowptr = d_orow; a.nbr = d_oth; a.wid = d_prow; a.eid = d_ein; a.eid2 = d_eout; a.wP = 2147483647;
  SpecArgs<float> b{};
  b.N = N; b.mul = mul; b.din = din; b.dout = dout; b.wn = wn;
  b.gxe = gxe; b.out = out; b.rowptr = d_trow; b.eid = d_tslot;

  const int64_t items = (int64_t)N * nchunk;
  auto launch_pair = [&]() {
#ifdef LAB_SRING
    {
      const int64_t witems = items * kPairParts;
      if (gxat) hipLaunchKernelGGL((bwd_pair_split_ring_kernel<true, true>), dim3((unsigned)((witems + 3) / 4)), dim3(256), (size_t)81920, 0, a);
      else hipLaunchKernelGGL((bwd_pair_split_ring_kernel<true, false>), dim3((unsigned)((witems + 3) / 4)), dim3(256), (size_t)81920, 0, a);
      return;
    }
#endif
#ifdef LAB_SPLIT
    const int64_t witems = items * kPairParts;
    hipLaunchKernelGGL((bwd_pair_split_kernel<float, true, true>), dim3((unsigned)((witems + 3) / 4)), dim3(256), 0, 0, a);
#else
#ifdef LAB_RING
    {
      const unsigned blocks = (unsigned)((items * wpn + 3) / 4);
      const size_t smem = (size_t)4 * kRingWaveBytes + extra_lds;
#define LAB_RING_GO(W) do { if (gxat) hipLaunchKernelGGL((bwd_pair_ring_kernel<W, true, true>), dim3(blocks), dim3(256), smem, 0, a); \
                         else hipLaunchKernelGGL((bwd_pair_ring_kernel<W, true, false>), dim3(blocks), dim3(256), smem, 0, a); } while (0)
      if (wpn >= 4) LAB_RING_GO(4); else if (wpn == 2) LAB_RING_GO(2); else LAB_RING_GO(1);
      return;
    }
#endif
#ifdef LAB_PK
#define LAB_PAIR_KERNEL bwd_pair_pk_kernel
#else
#define LAB_PAIR_KERNEL bwd_pair_kernel
#endif
    if (wpn >= 4) {
      hipLaunchKernelGGL((LAB_PAIR_KERNEL<float, 4, true, true>), dim3((unsigned)items), dim3(256), (size_t)3 * kXD * 64 * 4 + extra_lds, 0, a);
    } else {
      hipLaunchKernelGGL((LAB_PAIR_KERNEL<float, 1, true, true>), dim3((unsigned)((items + 3) / 4)), dim3(256), extra_lds, 0, a);
    }
#endif
  };
  auto launch_sum = [&]() {
    hipLaunchKernelGGL((gx_rows_sum_kernel<float, true>), dim3((unsigned)((items + 3) / 4)), dim3(256), 0, 0, b);
  };
#ifdef LAB_SRING
  CK(hipFuncSetAttribute((const void*)bwd_pair_split_ring_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920));
  CK(hipFuncSetAttribute((const void*)bwd_pair_split_ring_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920));
#endif
#ifdef LAB_RING
  const int ring_lds = 4 * kRingWaveBytes + (int)extra_lds;
#define LAB_RING_ATTR(W, AT) CK(hipFuncSetAttribute((const void*)bwd_pair_ring_kernel<W, true, AT>, hipFuncAttributeMaxDynamicSharedMemorySize, ring_lds))
  LAB_RING_ATTR(4, true); LAB_RING_ATTR(2, true); LAB_RING_ATTR(1, true); LAB_RING_ATTR(4, false); LAB_RING_ATTR(2, false); LAB_RING_ATTR(1, false);
#endif
  hipEvent_t e0, e1, e2;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  double tp = 0, ts = 0;
  for (int r = -3; r < reps; ++r) {
    CK(hipEventRecord(e0, 0));
    if (gxat) CK(hipMemsetAsync(gxe, 0, (gxat == 2 ? (size_t)8 : (size_t)1) * N * din * 4, 0));
    if (mode != 2) launch_pair();
    CK(hipEventRecord(e1, 0));
    if (mode != 1) launch_sum();
    CK(hipEventRecord(e2, 0));
    CK(hipEventSynchronize(e2));
    CK(hipGetLastError());
    float m1, m2;
    CK(hipEventElapsedTime(&m1, e0, e1));
    CK(hipEventElapsedTime(&m2, e1, e2));
    if (r >= 0) { tp += m1; ts += m2; }
  }
  // checksums after ONE clean evaluation (the row sum accumulates into `out`, which the pair kernel overwrites first)
  if (gxat) CK(hipMemsetAsync(gxe, 0, (gxat == 2 ? (size_t)8 : (size_t)1) * N * din * 4, 0));
  launch_pair();
  if (gxat == 2) {  // fold the eight per-XCD accumulators into the first (host; checksum only) and count the rows each one holds
    CK(hipDeviceSynchronize());
    std::vector<float> h((size_t)8 * N * din);
    CK(hipMemcpy(h.data(), gxe, h.size() * 4, hipMemcpyDeviceToHost));
    size_t touched = 0;
    for (int x = 0; x < 8; ++x)
      for (int n = 0; n < N; ++n) {
        bool t = false;
        for (int i = 0; i < din; ++i) {
          const float v = h[((size_t)x * N + n) * din + i];
          if (v != 0.f) t = true;
          if (x) h[(size_t)n * din + i] += v;
        }
        touched += t;
      }
    fprintf(stderr, "per-XCD accumulators: %zu (XCD, node) rows hold something = %.2f per node\n", touched, (double)touched / N);
    CK(hipMemcpy(gxe, h.data(), (size_t)N * din * 4, hipMemcpyHostToDevice));
  }
  launch_sum();
  CK(hipDeviceSynchronize());
  const double c_gw = checksum(gw, (size_t)P * wn), c_out = checksum(out, (size_t)N * din);
  double c_gy = 0;
  {
    std::vector<float> h((size_t)E * kS * gyn);
    CK(hipMemcpy(h.data(), gy, h.size() * 4, hipMemcpyDeviceToHost));
    for (int e = 0; e < E; ++e)
      for (int j = 0; j < kS; ++j) {
        double s = 0;
        for (int c = 0; c < gyn; ++c) s += h[((size_t)e * gyn + c) * kS + j];
        c_gy += s * ((e * 9 + j) % 5 + 1);
      }
  }
#ifdef LAB_TIMING
  {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t[8];
    CK(hipMemcpyToSymbol(HIP_SYMBOL(nqa_lab_tm), z, sizeof(z)));
    launch_pair();
    CK(hipDeviceSynchronize());
    CK(hipMemcpyFromSymbol(t, HIP_SYMBOL(nqa_lab_tm), sizeof(t)));
    const double n = (double)t[4];
    printf("timing per pair and wavefront (shader cycles, %.0f pairs): indices+issue %.0f | rows arrive %.0f | compute+store issue %.0f | "
           "stores drain %.0f | sum %.0f || per wavefront: loop start -> exit %.0f cycles, %.2f pairs\n", n, t[0] / n, t[1] / n, t[2] / n, t[3] / n,
           (t[0] + t[1] + t[2] + t[3]) / n, (double)t[5] / (double)t[6], n / (double)t[6]);
  }
#endif
  const double bytes_alg = (double)E * (8.0 * wn + 8.0 * kS + 16.0) + 4.0 * N * (2.0 * din + dout);
  printf("N=%d E=%d P=%d mul=%d relabel=%d wpn=%d | pair %.1f us  sum %.1f us  total %.1f us | alg %.3f GB -> %.2f TB/s "
         "(%.3f of 8) | chk gw %.6e gy %.6e gx %.6e\n",
         N, E, P, mul, relabel, wpn, tp / reps * 1e3, ts / reps * 1e3, (tp + ts) / reps * 1e3, bytes_alg / 1e9,
         bytes_alg / ((tp + ts) / reps * 1e-3) / 1e12, bytes_alg / ((tp + ts) / reps * 1e-3) / 8e12, c_gw, c_gy, c_out);
  return 0;
}
