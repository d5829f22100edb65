// Edge embedding kernels: real spherical harmonics + Bessel radial basis x polynomial cutoff, and their
// vector-Jacobian product (the edge -> position leg of the force backward).
//
// Replaces, for float64 edge vectors (nequip/nn/utils.py:68-118), the chain of small ATen ops behind
//   SphericalHarmonicEdgeAttrs.forward (nequip/nn/embedding/_edge.py:193-198),
//   EdgeLengthNormalizer.forward       (nequip/nn/embedding/_edge.py:65-80),
//   BesselEdgeLengthEncoding.forward   (nequip/nn/embedding/_edge.py:136-150),
//   PolynomialCutoff.forward           (nequip/nn/embedding/cutoffs.py:17-27),
//   ApplyFactor                        (nequip/model/nequip_models.py:318-322)
// with one pass over the edges: 24 B read, (S + nb) * sizeof(T) written per edge.  HBM-bound,
// one thread per edge (the per-edge work is a few hundred f64 FLOPs, the loads/stores are coalesced
// across the wave because consecutive threads handle consecutive edges).
// Arithmetic is float64 (the reference evaluates on float64 data and casts, _edge.py:140-142,196-197).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "generated/sh_generated.h"
#include "plan.h"

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
};

__device__ __forceinline__ double ipow_rt(double x, int n) {
  double r = 1.0, b = x;
  while (n > 0) {
    if (n & 1) r *= b;
    b *= b;
    n >>= 1;
  }
  return r;
}

__device__ __forceinline__ double pow_p(double x, double p, int p_int, int shift) {
  // x^(p + shift)
  if (p_int >= 0) {
    const int n = p_int + shift;
    return n >= 0 ? ipow_rt(x, n) : 1.0 / ipow_rt(x, -n);
  }
  return pow(x, p + (double)shift);
}

// cutoff(x) = 1 - (p+1)(p+2)/2 x^p + p(p+2) x^(p+1) - p(p+1)/2 x^(p+2), masked by x < 1 (cutoffs.py:23-27)
__device__ __forceinline__ double poly_cutoff(double x, double p, int p_int) {
  double out = 1.0;
  out = out - (((p + 1.0) * (p + 2.0) / 2.0) * pow_p(x, p, p_int, 0));
  out = out + (p * (p + 2.0) * pow_p(x, p, p_int, 1));
  out = out - ((p * (p + 1.0) / 2.0) * pow_p(x, p, p_int, 2));
  return x < 1.0 ? out : 0.0;
}

__device__ __forceinline__ double poly_cutoff_grad(double x, double p, int p_int) {
  if (!(x < 1.0)) return 0.0;
  double g = -(((p + 1.0) * (p + 2.0) / 2.0) * p * pow_p(x, p, p_int, -1));
  g += p * (p + 2.0) * (p + 1.0) * pow_p(x, p, p_int, 0);
  g -= (p * (p + 1.0) / 2.0) * (p + 2.0) * pow_p(x, p, p_int, 1);
  return g;
}

// torch.sinc: sin(pi t)/(pi t), 1 at t == 0
__device__ __forceinline__ double sinc_pi(double t) {
  if (t == 0.0) return 1.0;
  const double a = kPi * t;
  return sin(a) / a;
}

template <typename T, int L>
__global__ __launch_bounds__(256) void edge_embed_fwd_kernel(const EdgeEmbedParams prm, T* __restrict__ sh,
                                                             T* __restrict__ emb, T* __restrict__ cutoff) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= prm.E) return;
  constexpr int S = (L + 1) * (L + 1);
  const double vx = prm.vec[3 * e + 0], vy = prm.vec[3 * e + 1], vz = prm.vec[3 * e + 2];
  const double r = sqrt(vx * vx + vy * vy + vz * vz);
  if (sh != nullptr) {
    // torch.nn.functional.normalize: v / max(|v|, 1e-12)
    const double inv = 1.0 / fmax(r, 1e-12);
    double Y[S];
    nqa_sh_eval<L>(vx * inv, vy * inv, vz * inv, Y);
    T* __restrict__ o = sh + e * S;
#pragma unroll
    for (int s = 0; s < S; ++s) o[s] = (T)Y[s];
  }
  if (emb != nullptr || cutoff != nullptr) {
    const double rr = prm.rr_edge ? prm.rr_edge[e] : prm.rr;
    const double x = r * rr;
    const T c = (T)poly_cutoff(x, prm.p, prm.p_int);
    if (cutoff != nullptr) cutoff[e] = c;
    if (emb != nullptr) {
      const T f = (T)prm.factor;
      T* __restrict__ o = emb + e * prm.nb;
      for (int n = 0; n < prm.nb; ++n) {
        const double w = prm.bw[n];
        const T b = (T)(sinc_pi(x * w) * w);
        o[n] = f * (b * c);
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
  constexpr int S = (L + 1) * (L + 1);
  const double vx = prm.vec[3 * e + 0], vy = prm.vec[3 * e + 1], vz = prm.vec[3 * e + 2];
  const double r = sqrt(vx * vx + vy * vy + vz * vz);
  const double inv = 1.0 / fmax(r, 1e-12);
  const double ux = vx * inv, uy = vy * inv, uz = vz * inv;
  double gx = 0.0, gy = 0.0, gz = 0.0;
  if (g_sh != nullptr) {
    double g[S];
    const T* __restrict__ gi = g_sh + e * S;
#pragma unroll
    for (int s = 0; s < S; ++s) g[s] = (double)gi[s];
    double G[3];
    nqa_sh_vjp<L>(ux, uy, uz, g, G);
    if (r >= 1e-12) {
      // d(v/|v|)/dv = (I - u u^T)/|v|
      const double ug = ux * G[0] + uy * G[1] + uz * G[2];
      gx = (G[0] - ux * ug) * inv;
      gy = (G[1] - uy * ug) * inv;
      gz = (G[2] - uz * ug) * inv;
    } else {
      // clamped branch of normalize: v * 1e12
      gx = G[0] * inv;
      gy = G[1] * inv;
      gz = G[2] * inv;
    }
  }
  if (g_emb != nullptr) {
    const double rr = prm.rr_edge ? prm.rr_edge[e] : prm.rr;
    const double x = r * rr;
    const double c = poly_cutoff(x, prm.p, prm.p_int);
    const double dc = poly_cutoff_grad(x, prm.p, prm.p_int);
    const T* __restrict__ gi = g_emb + e * prm.nb;
    double acc = 0.0;
    for (int n = 0; n < prm.nb; ++n) {
      const double w = prm.bw[n];
      const double t = x * w;
      const double sc = sinc_pi(t);
      // d/dx [sinc(x w) w] = w^2 (cos(pi t) - sinc(t)) / t
      const double db = (t == 0.0) ? 0.0 : w * w * (cos(kPi * t) - sc) / t;
      acc += (double)gi[n] * (db * c + sc * w * dc);
    }
    const double gr = acc * prm.factor * rr;  // dE/dr
    // d|v|/dv = u (sqrt backward; 0-length edges do not occur in neighbour lists)
    gx += gr * ux;
    gy += gr * uy;
    gz += gr * uz;
  }
  g_vec[3 * e + 0] = gx;
  g_vec[3 * e + 1] = gy;
  g_vec[3 * e + 2] = gz;
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

template <typename T>
static int launch_fwd(int lmax, const EdgeEmbedParams& prm, void* sh, void* emb, void* cutoff, hipStream_t s) {
  if (prm.E == 0) return NQA_OK;
  const unsigned grid = (unsigned)((prm.E + 255) / 256);
#define NQA_LAUNCH(L)                                                                                      \
  hipLaunchKernelGGL((edge_embed_fwd_kernel<T, L>), dim3(grid), dim3(256), 0, s, prm, static_cast<T*>(sh), \
                     static_cast<T*>(emb), static_cast<T*>(cutoff))
  switch (lmax) {
    case 0: NQA_LAUNCH(0); break;
    case 1: NQA_LAUNCH(1); break;
    case 2: NQA_LAUNCH(2); break;
    case 3: NQA_LAUNCH(3); break;
    case 4: NQA_LAUNCH(4); break;
    default: set_error("nqa_edge_embed_fwd: lmax exceeds supported maximum"); return NQA_ERR_UNSUPPORTED;
  }
#undef NQA_LAUNCH
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_error(std::string("nqa_edge_embed_fwd: ") + hipGetErrorString(err));
    return NQA_ERR_LAUNCH;
  }
  return NQA_OK;
}

template <typename T>
static int launch_bwd(int lmax, const EdgeEmbedParams& prm, const void* g_sh, const void* g_emb, double* g_vec,
                      hipStream_t s) {
  if (prm.E == 0) return NQA_OK;
  const unsigned grid = (unsigned)((prm.E + 255) / 256);
#define NQA_LAUNCH(L)                                                                                    \
  hipLaunchKernelGGL((edge_embed_bwd_kernel<T, L>), dim3(grid), dim3(256), 0, s, prm,                    \
                     static_cast<const T*>(g_sh), static_cast<const T*>(g_emb), g_vec)
  switch (lmax) {
    case 0: NQA_LAUNCH(0); break;
    case 1: NQA_LAUNCH(1); break;
    case 2: NQA_LAUNCH(2); break;
    case 3: NQA_LAUNCH(3); break;
    case 4: NQA_LAUNCH(4); break;
    default: set_error("nqa_edge_embed_bwd: lmax exceeds supported maximum"); return NQA_ERR_UNSUPPORTED;
  }
#undef NQA_LAUNCH
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_error(std::string("nqa_edge_embed_bwd: ") + hipGetErrorString(err));
    return NQA_ERR_LAUNCH;
  }
  return NQA_OK;
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

}  // extern "C"
