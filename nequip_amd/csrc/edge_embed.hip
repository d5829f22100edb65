// Edge embedding kernels: real spherical harmonics + Bessel radial basis x polynomial cutoff, their
// vector-Jacobian product (the edge -> position leg of the force backward) and the derivative of that product
// (second order, needed when forces enter a training loss).
//
// Replaces, for float64 edge vectors (nequip/nn/utils.py:68-118), the chain of small ATen ops behind
//   SphericalHarmonicEdgeAttrs.forward (nequip/nn/embedding/_edge.py:193-198),
//   EdgeLengthNormalizer.forward       (nequip/nn/embedding/_edge.py:65-80),
//   BesselEdgeLengthEncoding.forward   (nequip/nn/embedding/_edge.py:136-150),
//   PolynomialCutoff.forward           (nequip/nn/embedding/cutoffs.py:17-27),
//   ApplyFactor                        (nequip/model/nequip_models.py:318-322)
// with one pass over the edges: 24 B read, (S + nb) * sizeof(T) written per edge.  HBM-bound, one thread per
// edge (a few hundred f64 FLOPs each; consecutive threads handle consecutive edges).
// Arithmetic is float64 (the reference evaluates on float64 data and casts, _edge.py:140-142,196-197).
//
// The per-edge maps are written once, templated on the scalar type S: S = double gives the forward and the VJP;
// S = Dual (forward-mode dual number seeded with the cotangent of the VJP output) gives, from the very same code,
// the Jacobian-vector product J c and the Hessian contraction (sum_k g_k H_k) c that the double backward of
// ForceStressOutput (nequip/nn/grad_output.py:217-221 with create_graph=True) needs.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "plan.h"

namespace nqa {

// ---- forward-mode dual numbers --------------------------------------------------------------------------------
struct Dual {
  double v, d;
  __device__ __forceinline__ Dual() : v(0.0), d(0.0) {}
  __device__ __forceinline__ Dual(double a) : v(a), d(0.0) {}
  __device__ __forceinline__ Dual(double a, double b) : v(a), d(b) {}
};
__device__ __forceinline__ Dual operator+(const Dual& a, const Dual& b) { return Dual(a.v + b.v, a.d + b.d); }
__device__ __forceinline__ Dual operator-(const Dual& a, const Dual& b) { return Dual(a.v - b.v, a.d - b.d); }
__device__ __forceinline__ Dual operator-(const Dual& a) { return Dual(-a.v, -a.d); }
__device__ __forceinline__ Dual operator*(const Dual& a, const Dual& b) { return Dual(a.v * b.v, a.v * b.d + a.d * b.v); }
__device__ __forceinline__ Dual operator/(const Dual& a, const Dual& b) {
  const double q = a.v / b.v;
  return Dual(q, (a.d - q * b.d) / b.v);
}
__device__ __forceinline__ Dual operator+(const Dual& a, double b) { return Dual(a.v + b, a.d); }
__device__ __forceinline__ Dual operator+(double a, const Dual& b) { return Dual(a + b.v, b.d); }
__device__ __forceinline__ Dual operator-(const Dual& a, double b) { return Dual(a.v - b, a.d); }
__device__ __forceinline__ Dual operator-(double a, const Dual& b) { return Dual(a - b.v, -b.d); }
__device__ __forceinline__ Dual operator*(const Dual& a, double b) { return Dual(a.v * b, a.d * b); }
__device__ __forceinline__ Dual operator*(double a, const Dual& b) { return Dual(a * b.v, a * b.d); }
__device__ __forceinline__ Dual operator/(const Dual& a, double b) { return Dual(a.v / b, a.d / b); }
__device__ __forceinline__ Dual operator/(double a, const Dual& b) { return Dual(a) / b; }
__device__ __forceinline__ Dual& operator+=(Dual& a, const Dual& b) { a.v += b.v; a.d += b.d; return a; }
__device__ __forceinline__ Dual dsqrt(const Dual& a) {
  const double s = sqrt(a.v);
  return Dual(s, s > 0.0 ? 0.5 * a.d / s : 0.0);
}
__device__ __forceinline__ Dual dsin(const Dual& a) { return Dual(sin(a.v), cos(a.v) * a.d); }
__device__ __forceinline__ Dual dcos(const Dual& a) { return Dual(cos(a.v), -sin(a.v) * a.d); }
__device__ __forceinline__ double dsqrt(double a) { return sqrt(a); }
__device__ __forceinline__ double dsin(double a) { return sin(a); }
__device__ __forceinline__ double dcos(double a) { return cos(a); }
__device__ __forceinline__ double val(double a) { return a; }
__device__ __forceinline__ double val(const Dual& a) { return a.v; }
__device__ __forceinline__ double tan_part(double) { return 0.0; }
__device__ __forceinline__ double tan_part(const Dual& a) { return a.d; }

}  // namespace nqa

using nqa::Dual;
#include "generated/sh_generated.h"

namespace nqa {

constexpr int kMaxBessel = 32;
constexpr double kPi = 3.14159265358979323846;

struct EdgeEmbedParams {
  const double* __restrict__ vec;
  const double* __restrict__ rr_edge;  // optional per-edge 1/r_max
  const double* __restrict__ bw;       // bessel weights [nb]
  int64_t E;
  double rr;
  double p;
  double factor;
  int32_t nb;
  int32_t p_int;  // p as integer if integral, else -1
  // paired radial rows (nqa_edge_embed_*_paired): pair_row[e] < P for the representative edge of a pair (its row in the per-pair
  // arrays), >= P for the reverse edge.  emb_pairs [P, nb] / its cotangent are the per-pair forms of the radial embedding.
  const int32_t* __restrict__ pair_row;
  void* emb_pairs;
  const void* g_emb_pairs;
  int32_t P;
};

template <typename S>
__device__ __forceinline__ S ipow_rt(const S& x, int n) {
  S r = S(1.0), b = x;
  while (n > 0) {
    if (n & 1) r = r * b;
    b = b * b;
    n >>= 1;
  }
  return r;
}

__device__ __forceinline__ double pow_p(double x, double p, int p_int, int shift) {
  if (p_int >= 0) {
    const int n = p_int + shift;
    return n >= 0 ? ipow_rt<double>(x, n) : 1.0 / ipow_rt<double>(x, -n);
  }
  return pow(x, p + (double)shift);
}
__device__ __forceinline__ Dual pow_p(const Dual& x, double p, int p_int, int shift) {
  // d/dx x^q = q x^(q-1)
  const double q = (p_int >= 0 ? (double)p_int : p) + (double)shift;
  const double v = pow_p(x.v, p, p_int, shift);
  const double dv = q == 0.0 ? 0.0 : q * pow_p(x.v, p, p_int, shift - 1);
  return Dual(v, dv * x.d);
}

// cutoff(x) = 1 - (p+1)(p+2)/2 x^p + p(p+2) x^(p+1) - p(p+1)/2 x^(p+2), masked by x < 1 (cutoffs.py:23-27)
template <typename S>
__device__ __forceinline__ S poly_cutoff(const S& x, double p, int p_int) {
  if (!(val(x) < 1.0)) return S(0.0);
  S out = S(1.0);
  out = out - (((p + 1.0) * (p + 2.0) / 2.0) * pow_p(x, p, p_int, 0));
  out = out + (p * (p + 2.0) * pow_p(x, p, p_int, 1));
  out = out - ((p * (p + 1.0) / 2.0) * pow_p(x, p, p_int, 2));
  return out;
}
template <typename S>
__device__ __forceinline__ S poly_cutoff_grad(const S& x, double p, int p_int) {
  if (!(val(x) < 1.0)) return S(0.0);
  S g = -(((p + 1.0) * (p + 2.0) / 2.0) * p * pow_p(x, p, p_int, -1));
  g = g + p * (p + 2.0) * (p + 1.0) * pow_p(x, p, p_int, 0);
  g = g - (p * (p + 1.0) / 2.0) * (p + 2.0) * pow_p(x, p, p_int, 1);
  return g;
}

// ---- per-edge maps, templated on the scalar type ---------------------------------------------------------------
// geometry shared by all maps: r = |v|, u = v / max(r, 1e-12)  (torch.nn.functional.normalize)
template <typename S>
struct Geom {
  S r, inv, ux, uy, uz;
  bool clamped;
};
template <typename S>
__device__ __forceinline__ Geom<S> geom(const S& vx, const S& vy, const S& vz) {
  Geom<S> g;
  g.r = dsqrt(vx * vx + vy * vy + vz * vz);
  g.clamped = !(val(g.r) >= 1e-12);
  g.inv = g.clamped ? S(1e12) : 1.0 / g.r;
  g.ux = vx * g.inv;
  g.uy = vy * g.inv;
  g.uz = vz * g.inv;
  return g;
}

// Radial basis b_n(x) = w_n sinc(x w_n) (torch.sinc: sin(pi t)/(pi t)), x = r * rr, evaluated without a division or a
// transcendental per basis function:
//   b_n(x)  = sin(pi x w_n) / (pi x)            (the w_n of the sinc argument cancels),   b_n(0)  = w_n
//   b_n'(x) = (w_n cos(pi x w_n) - b_n(x)) / x                                           b_n'(0) = 0
// and, when the weights are the untrained Bessel roots w_n = n + 1 (checked on the device, wave-uniform), sin/cos of
// (n + 1) pi x follow from ONE sin/cos pair by angle addition (error grows ~ n ulp in f64; the outputs are cast to T).
template <typename S>
struct BesselBasis {
  S inv_x, inv_pix, s1, c1, s, c;
  bool zero, harmonic;
  __device__ __forceinline__ BesselBasis(const EdgeEmbedParams& prm, const S& x) {
    zero = val(x) == 0.0;
    inv_x = zero ? S(0.0) : 1.0 / x;
    inv_pix = inv_x * (1.0 / kPi);
    harmonic = true;
    for (int n = 0; n < prm.nb; ++n) harmonic = harmonic && (prm.bw[n] == (double)(n + 1));
    const S a = kPi * x;
    s1 = dsin(a);
    c1 = dcos(a);
    s = S(0.0);
    c = S(1.0);
  }
  // advance to basis function n (call with n = 0, 1, 2, ... in order); afterwards s, c = sin, cos(pi x w_n)
  __device__ __forceinline__ void step(const S& x, double w) {
    if (harmonic) {
      const S ns = s * c1 + c * s1;
      const S nc = c * c1 - s * s1;
      s = ns;
      c = nc;
    } else {
      const S a = kPi * (x * w);
      s = dsin(a);
      c = dcos(a);
    }
  }
  __device__ __forceinline__ S value(double w) const { return zero ? S(w) : s * inv_pix; }
  __device__ __forceinline__ S grad(double w) const { return zero ? S(0.0) : (w * c - s * inv_pix) * inv_x; }
};

// VJP of (sh, emb) w.r.t. the edge vector for cotangents (g_sh, g_emb); returns the three components in S.
template <typename T, int L, typename S>
__device__ __forceinline__ void embed_vjp(const EdgeEmbedParams& prm, int64_t e, const S& vx, const S& vy, const S& vz,
                                          const T* __restrict__ g_sh, const T* __restrict__ g_emb, S* out3) {
  constexpr int NS = (L + 1) * (L + 1);
  const Geom<S> gm = geom(vx, vy, vz);
  S gx = S(0.0), gy = S(0.0), gz = S(0.0);
  if (g_sh != nullptr) {
    double g[NS];
    const T* __restrict__ gi = g_sh + e * NS;
#pragma unroll
    for (int s = 0; s < NS; ++s) g[s] = (double)gi[s];
    S G[3];
    nqa_sh<L, S>::vjp(gm.ux, gm.uy, gm.uz, g, G);
    if (!gm.clamped) {
      // d(v/|v|)/dv = (I - u u^T)/|v|
      const S ug = gm.ux * G[0] + gm.uy * G[1] + gm.uz * G[2];
      gx = (G[0] - gm.ux * ug) * gm.inv;
      gy = (G[1] - gm.uy * ug) * gm.inv;
      gz = (G[2] - gm.uz * ug) * gm.inv;
    } else {
      gx = G[0] * gm.inv;
      gy = G[1] * gm.inv;
      gz = G[2] * gm.inv;
    }
  }
  int32_t prow = 0;
  const bool from_pairs = prm.g_emb_pairs != nullptr && (prow = prm.pair_row[e]) < prm.P;
  if (g_emb != nullptr || from_pairs) {
    const double rr = prm.rr_edge ? prm.rr_edge[e] : prm.rr;
    const S x = gm.r * rr;
    const S c = poly_cutoff(x, prm.p, prm.p_int);
    const S dc = poly_cutoff_grad(x, prm.p, prm.p_int);
    const T* __restrict__ gi = g_emb != nullptr ? g_emb + e * prm.nb : nullptr;
    const T* __restrict__ gp = from_pairs ? static_cast<const T*>(prm.g_emb_pairs) + (int64_t)prow * prm.nb : nullptr;
    S acc = S(0.0);
    BesselBasis<S> bb(prm, x);
    for (int n = 0; n < prm.nb; ++n) {
      const double w = prm.bw[n];
      bb.step(x, w);
      // (cotangent of the per-edge rows + cotangent of the per-pair row this edge represents)
      const double gn = (gi != nullptr ? (double)gi[n] : 0.0) + (gp != nullptr ? (double)gp[n] : 0.0);
      acc += gn * (bb.grad(w) * c + bb.value(w) * dc);
    }
    const S gr = acc * (prm.factor * rr);  // dE/dr ; d|v|/dv = u
    gx += gr * gm.ux;
    gy += gr * gm.uy;
    gz += gr * gm.uz;
  }
  out3[0] = gx;
  out3[1] = gy;
  out3[2] = gz;
}

template <typename T, int L>
__global__ __launch_bounds__(256) void edge_embed_fwd_kernel(const EdgeEmbedParams prm, T* __restrict__ sh,
                                                             T* __restrict__ emb, T* __restrict__ cutoff) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= prm.E) return;
  constexpr int NS = (L + 1) * (L + 1);
  const Geom<double> gm = geom<double>(prm.vec[3 * e + 0], prm.vec[3 * e + 1], prm.vec[3 * e + 2]);
  if (sh != nullptr) {
    double Y[NS];
    nqa_sh<L, double>::eval(gm.ux, gm.uy, gm.uz, Y);
    T* __restrict__ o = sh + e * NS;
#pragma unroll
    for (int s = 0; s < NS; ++s) o[s] = (T)Y[s];
  }
  if (emb != nullptr || cutoff != nullptr || prm.emb_pairs != nullptr) {
    const double rr = prm.rr_edge ? prm.rr_edge[e] : prm.rr;
    const double x = gm.r * rr;
    const T c = (T)poly_cutoff<double>(x, prm.p, prm.p_int);
    if (cutoff != nullptr) cutoff[e] = c;
    int32_t prow = 0;
    const bool to_pairs = prm.emb_pairs != nullptr && (prow = prm.pair_row[e]) < prm.P;
    if (emb != nullptr || to_pairs) {
      const T f = (T)prm.factor;
      T* __restrict__ o = emb != nullptr ? emb + e * prm.nb : nullptr;
      T* __restrict__ op = to_pairs ? static_cast<T*>(prm.emb_pairs) + (int64_t)prow * prm.nb : nullptr;
      BesselBasis<double> bb(prm, x);
      for (int n = 0; n < prm.nb; ++n) {
        const double w = prm.bw[n];
        bb.step(x, w);
        const T b = (T)bb.value(w);
        const T v = f * (b * c);  // same rounding order as the reference: factor * (bessel.to(T) * cutoff.to(T))
        if (o != nullptr) o[n] = v;
        if (op != nullptr) op[n] = v;
      }
    }
  }
}

template <typename T, int L>
__global__ __launch_bounds__(256) void edge_embed_bwd_kernel(const EdgeEmbedParams prm, const T* __restrict__ g_sh,
                                                             const T* __restrict__ g_emb,
                                                             double* __restrict__ g_vec) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= prm.E) return;
  double o[3];
  embed_vjp<T, L, double>(prm, e, prm.vec[3 * e + 0], prm.vec[3 * e + 1], prm.vec[3 * e + 2], g_sh, g_emb, o);
  g_vec[3 * e + 0] = o[0];
  g_vec[3 * e + 1] = o[1];
  g_vec[3 * e + 2] = o[2];
}

// Second order: given the cotangent c[E,3] of the VJP output g_vec = J(v)^T g,
//   gg_sh / gg_emb = J(v) c          (gradient w.r.t. the first-order cotangents g_sh / g_emb)
//   g_vec2         = (sum_k g_k H_k(v)) c   (gradient w.r.t. the edge vector)
template <typename T, int L>
__global__ __launch_bounds__(256) void edge_embed_bwd_bwd_kernel(const EdgeEmbedParams prm,
                                                                 const T* __restrict__ g_sh,
                                                                 const T* __restrict__ g_emb,
                                                                 const double* __restrict__ c, T* __restrict__ gg_sh,
                                                                 T* __restrict__ gg_emb,
                                                                 double* __restrict__ g_vec2) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= prm.E) return;
  constexpr int NS = (L + 1) * (L + 1);
  const Dual vx(prm.vec[3 * e + 0], c[3 * e + 0]), vy(prm.vec[3 * e + 1], c[3 * e + 1]),
      vz(prm.vec[3 * e + 2], c[3 * e + 2]);
  if (g_vec2 != nullptr) {
    Dual o[3];
    embed_vjp<T, L, Dual>(prm, e, vx, vy, vz, g_sh, g_emb, o);
    g_vec2[3 * e + 0] = o[0].d;
    g_vec2[3 * e + 1] = o[1].d;
    g_vec2[3 * e + 2] = o[2].d;
  }
  const Geom<Dual> gm = geom<Dual>(vx, vy, vz);
  if (gg_sh != nullptr) {
    Dual Y[NS];
    nqa_sh<L, Dual>::eval(gm.ux, gm.uy, gm.uz, Y);
    T* __restrict__ o = gg_sh + e * NS;
#pragma unroll
    for (int s = 0; s < NS; ++s) o[s] = (T)Y[s].d;
  }
  if (gg_emb != nullptr) {
    const double rr = prm.rr_edge ? prm.rr_edge[e] : prm.rr;
    const Dual x = gm.r * rr;
    const Dual cf = poly_cutoff<Dual>(x, prm.p, prm.p_int);
    T* __restrict__ o = gg_emb + e * prm.nb;
    BesselBasis<Dual> bb(prm, x);
    for (int n = 0; n < prm.nb; ++n) {
      const double w = prm.bw[n];
      bb.step(x, w);
      const Dual b = bb.value(w);
      o[n] = (T)(prm.factor * (b.d * cf.v + b.v * cf.d));
    }
  }
}

static int make_params(EdgeEmbedParams& prm, const double* edge_vec, int64_t E, double rmax_recip,
                       const double* rmax_recip_edge, int32_t nb, const double* bw, double p, double factor,
                       bool need_radial, const char* fn) {
  if (E < 0 || (E > 0 && edge_vec == nullptr)) {
    set_error(std::string(fn) + ": invalid edge vectors");
    return NQA_ERR_INVALID;
  }
  if (need_radial) {
    if (nb < 0 || nb > kMaxBessel || (nb > 0 && bw == nullptr)) {
      set_error(std::string(fn) + ": invalid bessel basis");
      return NQA_ERR_INVALID;
    }
    if (!(p >= 2.0)) {
      set_error(std::string(fn) + ": polynomial cutoff needs p >= 2");
      return NQA_ERR_INVALID;
    }
  }
  prm.vec = edge_vec;
  prm.rr_edge = rmax_recip_edge;
  prm.bw = bw;
  prm.E = E;
  prm.rr = rmax_recip;
  prm.p = p;
  prm.factor = factor;
  prm.nb = nb;
  const double pr = (double)(int)p;
  prm.p_int = (pr == p && p < 64.0) ? (int)p : -1;
  return NQA_OK;
}

static int finish(const char* fn) {
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_error(std::string(fn) + ": " + hipGetErrorString(err));
    return NQA_ERR_LAUNCH;
  }
  return NQA_OK;
}

#define NQA_SWITCH_L(lmax, LAUNCH, fn)                                                  \
  switch (lmax) {                                                                       \
    case 0: LAUNCH(0); break;                                                           \
    case 1: LAUNCH(1); break;                                                           \
    case 2: LAUNCH(2); break;                                                           \
    case 3: LAUNCH(3); break;                                                           \
    case 4: LAUNCH(4); break;                                                           \
    default: set_error(std::string(fn) + ": lmax exceeds supported maximum"); return NQA_ERR_UNSUPPORTED; \
  }

template <typename T>
static int launch_fwd(int lmax, const EdgeEmbedParams& prm, void* sh, void* emb, void* cutoff, hipStream_t s) {
  if (prm.E == 0) return NQA_OK;
  const unsigned grid = (unsigned)((prm.E + 255) / 256);
#define NQA_LAUNCH(L)                                                                                      \
  hipLaunchKernelGGL((edge_embed_fwd_kernel<T, L>), dim3(grid), dim3(256), 0, s, prm, static_cast<T*>(sh), \
                     static_cast<T*>(emb), static_cast<T*>(cutoff))
  NQA_SWITCH_L(lmax, NQA_LAUNCH, "nqa_edge_embed_fwd")
#undef NQA_LAUNCH
  return finish("nqa_edge_embed_fwd");
}

template <typename T>
static int launch_bwd(int lmax, const EdgeEmbedParams& prm, const void* g_sh, const void* g_emb, double* g_vec,
                      hipStream_t s) {
  if (prm.E == 0) return NQA_OK;
  const unsigned grid = (unsigned)((prm.E + 255) / 256);
#define NQA_LAUNCH(L)                                                                 \
  hipLaunchKernelGGL((edge_embed_bwd_kernel<T, L>), dim3(grid), dim3(256), 0, s, prm, \
                     static_cast<const T*>(g_sh), static_cast<const T*>(g_emb), g_vec)
  NQA_SWITCH_L(lmax, NQA_LAUNCH, "nqa_edge_embed_bwd")
#undef NQA_LAUNCH
  return finish("nqa_edge_embed_bwd");
}

template <typename T>
static int launch_bwd_bwd(int lmax, const EdgeEmbedParams& prm, const void* g_sh, const void* g_emb, const double* c,
                          void* gg_sh, void* gg_emb, double* g_vec2, hipStream_t s) {
  if (prm.E == 0) return NQA_OK;
  const unsigned grid = (unsigned)((prm.E + 255) / 256);
#define NQA_LAUNCH(L)                                                                                         \
  hipLaunchKernelGGL((edge_embed_bwd_bwd_kernel<T, L>), dim3(grid), dim3(256), 0, s, prm,                     \
                     static_cast<const T*>(g_sh), static_cast<const T*>(g_emb), c, static_cast<T*>(gg_sh),    \
                     static_cast<T*>(gg_emb), g_vec2)
  NQA_SWITCH_L(lmax, NQA_LAUNCH, "nqa_edge_embed_bwd_bwd")
#undef NQA_LAUNCH
  return finish("nqa_edge_embed_bwd_bwd");
}

}  // namespace nqa

using namespace nqa;

extern "C" {

int nqa_sh_lmax(void) { return NQA_SH_LMAX; }

int nqa_edge_embed_fwd(int32_t dtype, int32_t lmax, const double* edge_vec, int64_t num_edges, double rmax_recip,
                       const double* rmax_recip_edge, int32_t num_bessels, const double* bessel_weights,
                       double cutoff_p, double factor, void* sh, void* emb, void* cutoff, nqa_stream stream) {
  if (dtype != NQA_F32 && dtype != NQA_F64) {
    set_error("nqa_edge_embed_fwd: unsupported dtype");
    return NQA_ERR_UNSUPPORTED;
  }
  if (lmax < 0) {
    set_error("nqa_edge_embed_fwd: negative lmax");
    return NQA_ERR_INVALID;
  }
  EdgeEmbedParams prm{};
  int rc = make_params(prm, edge_vec, num_edges, rmax_recip, rmax_recip_edge, num_bessels, bessel_weights, cutoff_p,
                       factor, emb != nullptr || cutoff != nullptr, "nqa_edge_embed_fwd");
  if (rc != NQA_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == NQA_F32 ? launch_fwd<float>(lmax, prm, sh, emb, cutoff, s)
                          : launch_fwd<double>(lmax, prm, sh, emb, cutoff, s);
}

int nqa_edge_embed_bwd(int32_t dtype, int32_t lmax, const double* edge_vec, int64_t num_edges, double rmax_recip,
                       const double* rmax_recip_edge, int32_t num_bessels, const double* bessel_weights,
                       double cutoff_p, double factor, const void* g_sh, const void* g_emb, double* g_edge_vec,
                       nqa_stream stream) {
  if (dtype != NQA_F32 && dtype != NQA_F64) {
    set_error("nqa_edge_embed_bwd: unsupported dtype");
    return NQA_ERR_UNSUPPORTED;
  }
  if (lmax < 0 || (num_edges > 0 && g_edge_vec == nullptr)) {
    set_error("nqa_edge_embed_bwd: invalid argument");
    return NQA_ERR_INVALID;
  }
  EdgeEmbedParams prm{};
  int rc = make_params(prm, edge_vec, num_edges, rmax_recip, rmax_recip_edge, num_bessels, bessel_weights, cutoff_p,
                       factor, g_emb != nullptr, "nqa_edge_embed_bwd");
  if (rc != NQA_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == NQA_F32 ? launch_bwd<float>(lmax, prm, g_sh, g_emb, g_edge_vec, s)
                          : launch_bwd<double>(lmax, prm, g_sh, g_emb, g_edge_vec, s);
}

int nqa_edge_embed_fwd_paired(int32_t dtype, int32_t lmax, const double* edge_vec, int64_t num_edges, double rmax_recip,
                              int32_t num_bessels, const double* bessel_weights, double cutoff_p, double factor,
                              const int32_t* pair_row, int64_t num_pairs, void* sh, void* emb, void* emb_pairs,
                              nqa_stream stream) {
  if (dtype != NQA_F32 && dtype != NQA_F64) {
    set_error("nqa_edge_embed_fwd_paired: unsupported dtype");
    return NQA_ERR_UNSUPPORTED;
  }
  if (lmax < 0 || num_pairs < 0 || num_pairs > 0x7fffffff || (num_edges > 0 && (pair_row == nullptr || emb_pairs == nullptr))) {
    set_error("nqa_edge_embed_fwd_paired: invalid argument");
    return NQA_ERR_INVALID;
  }
  EdgeEmbedParams prm{};
  int rc = make_params(prm, edge_vec, num_edges, rmax_recip, nullptr, num_bessels, bessel_weights, cutoff_p, factor, true,
                       "nqa_edge_embed_fwd_paired");
  if (rc != NQA_OK) return rc;
  prm.pair_row = pair_row;
  prm.emb_pairs = emb_pairs;
  prm.P = (int32_t)num_pairs;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == NQA_F32 ? launch_fwd<float>(lmax, prm, sh, emb, nullptr, s)
                          : launch_fwd<double>(lmax, prm, sh, emb, nullptr, s);
}

int nqa_edge_embed_bwd_paired(int32_t dtype, int32_t lmax, const double* edge_vec, int64_t num_edges, double rmax_recip,
                              int32_t num_bessels, const double* bessel_weights, double cutoff_p, double factor,
                              const int32_t* pair_row, int64_t num_pairs, const void* g_sh, const void* g_emb,
                              const void* g_emb_pairs, double* g_edge_vec, nqa_stream stream) {
  if (dtype != NQA_F32 && dtype != NQA_F64) {
    set_error("nqa_edge_embed_bwd_paired: unsupported dtype");
    return NQA_ERR_UNSUPPORTED;
  }
  if (lmax < 0 || num_pairs < 0 || num_pairs > 0x7fffffff || (num_edges > 0 && g_edge_vec == nullptr) ||
      (g_emb_pairs != nullptr && pair_row == nullptr)) {
    set_error("nqa_edge_embed_bwd_paired: invalid argument");
    return NQA_ERR_INVALID;
  }
  EdgeEmbedParams prm{};
  int rc = make_params(prm, edge_vec, num_edges, rmax_recip, nullptr, num_bessels, bessel_weights, cutoff_p, factor,
                       g_emb != nullptr || g_emb_pairs != nullptr, "nqa_edge_embed_bwd_paired");
  if (rc != NQA_OK) return rc;
  prm.pair_row = pair_row;
  prm.g_emb_pairs = g_emb_pairs;
  prm.P = (int32_t)num_pairs;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == NQA_F32 ? launch_bwd<float>(lmax, prm, g_sh, g_emb, g_edge_vec, s)
                          : launch_bwd<double>(lmax, prm, g_sh, g_emb, g_edge_vec, s);
}

int nqa_edge_embed_bwd_bwd(int32_t dtype, int32_t lmax, const double* edge_vec, int64_t num_edges, double rmax_recip,
                           const double* rmax_recip_edge, int32_t num_bessels, const double* bessel_weights,
                           double cutoff_p, double factor, const void* g_sh, const void* g_emb,
                           const double* cot_g_edge_vec, void* gg_sh, void* gg_emb, double* g_edge_vec2,
                           nqa_stream stream) {
  if (dtype != NQA_F32 && dtype != NQA_F64) {
    set_error("nqa_edge_embed_bwd_bwd: unsupported dtype");
    return NQA_ERR_UNSUPPORTED;
  }
  if (lmax < 0 || (num_edges > 0 && cot_g_edge_vec == nullptr) || (gg_sh != nullptr && g_sh == nullptr && false)) {
    set_error("nqa_edge_embed_bwd_bwd: invalid argument");
    return NQA_ERR_INVALID;
  }
  EdgeEmbedParams prm{};
  int rc = make_params(prm, edge_vec, num_edges, rmax_recip, rmax_recip_edge, num_bessels, bessel_weights, cutoff_p,
                       factor, g_emb != nullptr || gg_emb != nullptr, "nqa_edge_embed_bwd_bwd");
  if (rc != NQA_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == NQA_F32
             ? launch_bwd_bwd<float>(lmax, prm, g_sh, g_emb, cot_g_edge_vec, gg_sh, gg_emb, g_edge_vec2, s)
             : launch_bwd_bwd<double>(lmax, prm, g_sh, g_emb, cot_g_edge_vec, gg_sh, gg_emb, g_edge_vec2, s);
}

}  // extern "C"
