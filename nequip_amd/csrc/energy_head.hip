// Per-atom energy head in one launch per direction.
//
// After the last convolution the reference runs, on the N atom rows (nequip/model/nequip_models.py:371-399):
//   Gate (the last layer keeps scalars only: x = cst * act(h))                  nequip/nn/convnetlayer.py:162-164
//   ScalarMLP readout, depth 0: e = x @ (W * alpha), float32                    nequip/nn/mlp.py:262-268
//   PerTypeScaleShift: E_atom = shift[type] + scale[type] * double(e)           nequip/nn/atomwise.py:116-284
// and autograd runs the same chain backwards -- a dozen launches on [N, 64] / [N, 1] tensors that cost more in launch
// latency than in work.  Here: forward one launch (h -> E_atom, float64), backward one launch (dE_atom -> dh).
// Arithmetic as in the reference: the dot product in float32, scale / shift in float64.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "plan.h"

namespace nqa {

template <typename T>
__device__ __forceinline__ T eh_act(int act, T x, T cst) {
  if (act == 1) return cst * x / (T(1) + expf(-x));
  if (act == 2) return cst * tanhf(x);
  return x;
}
template <typename T>
__device__ __forceinline__ T eh_act_grad(int act, T x, T cst) {
  if (act == 1) {
    const T s = T(1) / (T(1) + expf(-x));
    return cst * s * (T(1) + x * (T(1) - s));
  }
  if (act == 2) {
    const T t = tanhf(x);
    return cst * (T(1) - t * t);
  }
  return T(1);
}

struct EnergyHeadArgs {
  const float* __restrict__ h;        // [N, D] pre-activation scalars
  const float* __restrict__ w;        // [D] readout weights (alpha folded in)
  const double* __restrict__ scales;  // [n_scales] or NULL
  const double* __restrict__ shifts;  // [n_shifts] or NULL
  const int64_t* __restrict__ types;  // [N] (needed when n_scales > 1 or n_shifts > 1)
  const double* __restrict__ g_e;     // backward: [N] gradient w.r.t. the per-atom energies
  double* __restrict__ e_atom;        // forward: [N]
  float* __restrict__ g_h;            // backward: [N, D]
  int64_t N;
  int32_t D, act, n_scales, n_shifts;
  float cst;
};

// 16 lanes per atom, four atoms per wavefront; a lane walks the row in float4 steps of 16 lanes (D % 4 == 0)
__global__ __launch_bounds__(256) void energy_head_fwd_kernel(const EnergyHeadArgs a) {
  const int sub = threadIdx.x & 15;
  const int64_t z = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
  const bool ok = z < a.N;
  const float* __restrict__ row = a.h + (ok ? z : 0) * a.D;
  float s = 0.f;
  for (int c = 4 * sub; c < a.D; c += 64) {
    const float4 hv = *reinterpret_cast<const float4*>(row + c);
    const float4 wv = *reinterpret_cast<const float4*>(a.w + c);
    s += wv.x * eh_act(a.act, hv.x, a.cst) + wv.y * eh_act(a.act, hv.y, a.cst) + wv.z * eh_act(a.act, hv.z, a.cst) +
         wv.w * eh_act(a.act, hv.w, a.cst);
  }
  s += __shfl_xor(s, 8);
  s += __shfl_xor(s, 4);
  s += __shfl_xor(s, 2);
  s += __shfl_xor(s, 1);
  if (ok && sub == 0) {
    const int t = (a.n_scales > 1 || a.n_shifts > 1) ? (int)a.types[z] : 0;
    double e = (double)s;
    if (a.scales != nullptr) e *= a.scales[a.n_scales > 1 ? t : 0];
    if (a.shifts != nullptr) e += a.shifts[a.n_shifts > 1 ? t : 0];
    a.e_atom[z] = e;
  }
}

__global__ __launch_bounds__(256) void energy_head_bwd_kernel(const EnergyHeadArgs a) {
  const int q = a.D >> 2;  // float4 per row
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= a.N * q) return;
  const int64_t z = idx / q;
  const int c = (int)(idx - z * q) * 4;
  double g = a.g_e[z];
  if (a.scales != nullptr) g *= a.scales[a.n_scales > 1 ? (int)a.types[z] : 0];
  const float gf = (float)g;
  const float4 hv = *reinterpret_cast<const float4*>(a.h + z * a.D + c);
  const float4 wv = *reinterpret_cast<const float4*>(a.w + c);
  float4 r;
  r.x = gf * wv.x * eh_act_grad(a.act, hv.x, a.cst);
  r.y = gf * wv.y * eh_act_grad(a.act, hv.y, a.cst);
  r.z = gf * wv.z * eh_act_grad(a.act, hv.z, a.cst);
  r.w = gf * wv.w * eh_act_grad(a.act, hv.w, a.cst);
  *reinterpret_cast<float4*>(a.g_h + z * a.D + c) = r;
}

}  // namespace nqa

extern "C" {

int nqa_energy_head(int32_t backward, const void* h, const void* readout_weight, const void* scales, int32_t n_scales,
                    const void* shifts, int32_t n_shifts, const int64_t* atom_types, const void* grad_e, void* out,
                    int32_t dim, int32_t act, double cst, int64_t num_nodes, nqa_stream stream) {
  using namespace nqa;
  if (num_nodes < 0 || dim <= 0 || (dim & 3) != 0 || act < 0 || act > 2 || n_scales < 0 || n_shifts < 0 ||
      (num_nodes > 0 && (!h || !readout_weight || !out || (backward && !grad_e))) ||
      ((n_scales > 1 || n_shifts > 1) && atom_types == nullptr) || (n_scales > 0 && !scales) || (n_shifts > 0 && !shifts)) {
    set_error("nqa_energy_head: invalid argument (dim must be a positive multiple of 4)");
    return NQA_ERR_INVALID;
  }
  if (num_nodes == 0) return NQA_OK;
  EnergyHeadArgs a{};
  a.h = static_cast<const float*>(h);
  a.w = static_cast<const float*>(readout_weight);
  a.scales = n_scales > 0 ? static_cast<const double*>(scales) : nullptr;
  a.shifts = n_shifts > 0 ? static_cast<const double*>(shifts) : nullptr;
  a.types = atom_types;
  a.N = num_nodes;
  a.D = dim;
  a.act = act;
  a.n_scales = n_scales;
  a.n_shifts = n_shifts;
  a.cst = (float)cst;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (backward) {
    a.g_e = static_cast<const double*>(grad_e);
    a.g_h = static_cast<float*>(out);
    const int64_t items = num_nodes * (dim >> 2);
    hipLaunchKernelGGL(energy_head_bwd_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, a);
  } else {
    a.e_atom = static_cast<double*>(out);
    hipLaunchKernelGGL(energy_head_fwd_kernel, dim3((unsigned)((num_nodes * 16 + 255) / 256)), dim3(256), 0, s, a);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error(std::string("nqa_energy_head: ") + hipGetErrorString(e));
    return NQA_ERR_LAUNCH;
  }
  return NQA_OK;
}

}  // extern "C"
