// Radial network on the matrix cores: edge_weight = silu(edge_embedding @ (W0*a0)) @ (W1*a1), and its
// vector-Jacobian product w.r.t. the edge embedding (the radial leg of the force backward).
//
// Replaces ScalarMLPFunction.forward as InteractionBlock.edge_mlp (nequip/nn/interaction_block.py:119-127,196;
// nequip/nn/mlp.py:141-156,194-196,262-268: bias-free layers y = x @ (W * alpha), alpha = gain/sqrt(fan_in),
// SiLU in between) for the standard one-hidden-layer radial MLP, and the autograd of those mm/SiLU ops.
//
// This is the one true dense GEMM on the hot path ([E,H] x [H,W], K = H = 64/128): it runs on exact-fp32 MFMA
// (v_mfma_f32_32x32x2_f32: bitwise an fp32 fma chain, 157 TFLOP/s peak) -- no reduced precision.
//   forward : a workgroup owns 128 edges; each wavefront keeps the SiLU-activated hidden rows of its 32 edges in
//             VGPRs as MFMA A-fragments for the whole kernel (the hidden layer never touches HBM) and streams
//             the second-layer weights through LDS in 64-column chunks;
//   backward: g_h = g_w @ (W1*a1)^T with K = W streamed from HBM through LDS in 32-column chunks, then
//             g_emb = (g_h * silu'(pre)) @ (W0*a0)^T with the pre-activations recomputed from the embedding
//             (8 FMAs per element) instead of being stored.
// Roofline: MFMA-bound (2*H*W FLOP per edge vs 4*W bytes written/read per edge).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "plan.h"

namespace nqa {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kMlpRows = 128;  // edges per workgroup (4 wavefronts x 32 rows)
constexpr int kMaxNb = 8;    // radial basis size limit of the fused kernels (nequip default num_bessels = 8)

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt (CDNA4 counts stores in
// vmcnt), which would make every chunk wait for its own HBM stores / prefetch loads at the barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float silu_grad_f(float x) {
  const float s = 1.0f / (1.0f + __expf(-x));
  return s * (1.0f + x * (1.0f - s));
}

// ------------------------------------------------------------------------------------------------------------
// forward: out[E, W] = silu(emb[E, NB] @ W0s[NB, H]) @ W1s[H, W]
// ------------------------------------------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__(256) void radial_mlp_fwd_kernel(const float* __restrict__ emb,
                                                             const float* __restrict__ W0,
                                                             const float* __restrict__ W1, float a0, float a1,
                                                             int nb, int W, int64_t E, float* __restrict__ out) {
  constexpr int BN = 64;            // output columns per chunk
  constexpr int KP = H / 2;         // MFMA k-pairs
  __shared__ float w0s[H * kMaxNb];  // [k][c] (c contiguous)
  __shared__ float bs[2][H * BN];    // double-buffered W1 chunk [k][n]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = tid >> 6;
  const int half = lane >> 5;
  const int l31 = lane & 31;
  const int64_t row0 = (int64_t)blockIdx.x * kMlpRows + wv * 32;

  for (int i = tid; i < H * kMaxNb; i += 256) {
    const int k = i / kMaxNb, c = i - k * kMaxNb;
    w0s[i] = c < nb ? W0[c * H + k] * a0 : 0.f;  // zero padding: the unrolled dot products run over kMaxNb
  }
  // first W1 chunk
  const int nchunks = (W + BN - 1) / BN;
  // global -> registers (issued early) and registers -> LDS (written late): the L2 latency of the next W1 chunk
  // hides under the MFMAs of the current one
  constexpr int NV = H * (BN / 4) / 256;  // float4 per thread per chunk
  float4 pre[NV];
  auto stage_load = [&](int n0) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int i = tid + v * 256;
      const int k = i / (BN / 4), q = i - k * (BN / 4);
      const int n = n0 + q * 4;
      float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n + 3 < W) t4 = *reinterpret_cast<const float4*>(W1 + (int64_t)k * W + n);  // W % 4 == 0
      pre[v] = t4;
    }
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int i = tid + v * 256;
      const int k = i / (BN / 4), q = i - k * (BN / 4);
      float4 t4 = pre[v];
      t4.x *= a1; t4.y *= a1; t4.z *= a1; t4.w *= a1;
      *reinterpret_cast<float4*>(&bs[buf][k * BN + q * 4]) = t4;
    }
  };
  stage_load(0);
  stage_store(0);
  __syncthreads();

  // hidden activations of this lane's row as A fragments: a[t] = h[row][2t + half]
  float a[KP];
  {
    const int64_t row = row0 + l31;
    float ev[kMaxNb];
#pragma unroll
    for (int c = 0; c < kMaxNb; ++c) ev[c] = (c < nb && row < E) ? emb[row * nb + c] : 0.f;
#pragma unroll
    for (int t = 0; t < KP; ++t) {
      const int k = 2 * t + half;
      float pre = 0.f;
#pragma unroll
      for (int c = 0; c < kMaxNb; ++c) pre += ev[c] * w0s[k * kMaxNb + c];
      a[t] = (row < E) ? silu_f(pre) : 0.f;
    }
  }

  for (int ch = 0; ch < nchunks; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < nchunks) stage_load((ch + 1) * BN);
    const float* __restrict__ b = bs[buf] + half * BN + l31;
    f32x16 acc0 = {0}, acc1 = {0};
    // B fragments are fetched from LDS one register batch (TB k-pairs) ahead of the MFMAs that consume them
    constexpr int TB = 8;
    float bq[2][TB][2];
#pragma unroll
    for (int i = 0; i < TB; ++i) {
      bq[0][i][0] = b[(2 * i) * BN];
      bq[0][i][1] = b[(2 * i) * BN + 32];
    }
#pragma unroll
    for (int tb = 0; tb < KP / TB; ++tb) {
      if (tb + 1 < KP / TB) {
#pragma unroll
        for (int i = 0; i < TB; ++i) {
          bq[(tb + 1) & 1][i][0] = b[(2 * ((tb + 1) * TB + i)) * BN];
          bq[(tb + 1) & 1][i][1] = b[(2 * ((tb + 1) * TB + i)) * BN + 32];
        }
      }
      // pin the order: the compiler otherwise sinks each LDS read next to its MFMA and waits lgkmcnt(0) per pair
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TB; ++i) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tb * TB + i], bq[tb & 1][i][0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tb * TB + i], bq[tb & 1][i][1], acc1, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // next chunk registers -> LDS first (its loads were issued before the MFMAs and have landed), then the output
    // stores of this chunk, which nothing below waits for
    if (ch + 1 < nchunks) stage_store(buf ^ 1);
    const int n0 = ch * BN;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (row < E) {
        const int c0 = n0 + l31, c1 = n0 + 32 + l31;
        if (c0 < W) out[row * W + c0] = acc0[r];
        if (c1 < W) out[row * W + c1] = acc1[r];
      }
    }
    lds_barrier();
  }
}

// ------------------------------------------------------------------------------------------------------------
// backward: g_emb[E, NB] = ((g_w[E, W] @ W1s^T[W, H]) * silu'(pre)) @ W0s^T[H, NB]
// ------------------------------------------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__(256) void radial_mlp_bwd_kernel(const float* __restrict__ emb,
                                                             const float* __restrict__ W0,
                                                             const float* __restrict__ W1,
                                                             const float* __restrict__ gw, float a0, float a1, int nb,
                                                             int W, int64_t E, float* __restrict__ g_emb) {
  constexpr int BK = 32;            // k (= output-column of the forward) per chunk
  constexpr int AS = BK + 1;        // padded row stride of the A tile (bank-conflict free column reads)
  constexpr int NT = H / 32;        // 32-wide column tiles of the hidden layer per wavefront
  constexpr int GS = H + 1;         // padded row stride of the g_pre tile
  // LDS: main loop uses as[2][128*AS] + bs[2][BK*H]; the epilogue re-uses the same memory for g_pre[128][GS]
  constexpr int kMain = 2 * (kMlpRows * AS + BK * H);
  constexpr int kEpi = kMlpRows * GS;
  constexpr int kBuf = kMain > kEpi ? kMain : kEpi;
  __shared__ float smem[kBuf];
  __shared__ float w0s[H * kMaxNb];       // [k][c]
  __shared__ float w0t[kMaxNb * H];       // [c][k]
  __shared__ float es[kMlpRows * kMaxNb];  // embedding tile [row][c]
  float* as0 = smem;
  float* bs0 = smem + 2 * kMlpRows * AS;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = tid >> 6;
  const int half = lane >> 5;
  const int l31 = lane & 31;
  const int64_t blk0 = (int64_t)blockIdx.x * kMlpRows;

  for (int i = tid; i < H * kMaxNb; i += 256) {
    const int k = i / kMaxNb, c = i - k * kMaxNb;
    const float v = c < nb ? W0[c * H + k] * a0 : 0.f;
    w0s[i] = v;
    w0t[c * H + k] = v;
  }
  for (int i = tid; i < kMlpRows * kMaxNb; i += 256) {
    const int r = i / kMaxNb, c = i - r * kMaxNb;
    es[i] = (c < nb && blk0 + r < E) ? emb[(blk0 + r) * nb + c] : 0.f;
  }

  const int nchunks = (W + BK - 1) / BK;
  constexpr int NA = kMlpRows * (BK / 4) / 256;  // float4 per thread: g_w tile
  constexpr int NB4 = H * (BK / 4) / 256;        // float4 per thread: W1 tile
  float4 pa[NA], pb[NB4];
  auto stage_load = [&](int k0) {
#pragma unroll
    for (int v = 0; v < NA; ++v) {
      const int i = tid + v * 256;
      const int r = i / (BK / 4), q = i - r * (BK / 4);
      const int64_t row = blk0 + r;
      const int k = k0 + q * 4;
      float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < E && k + 3 < W) t4 = *reinterpret_cast<const float4*>(gw + row * W + k);  // W % 4 == 0
      pa[v] = t4;
    }
#pragma unroll
    for (int v = 0; v < NB4; ++v) {
      const int i = tid + v * 256;
      const int n = i / (BK / 4), q = i - n * (BK / 4);
      const int k = k0 + q * 4;
      float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k + 3 < W) t4 = *reinterpret_cast<const float4*>(W1 + (int64_t)n * W + k);
      pb[v] = t4;
    }
  };
  auto stage_store = [&](int buf) {
    float* __restrict__ as = as0 + buf * kMlpRows * AS;
    float* __restrict__ bs = bs0 + buf * BK * H;
#pragma unroll
    for (int v = 0; v < NA; ++v) {
      const int i = tid + v * 256;
      const int r = i / (BK / 4), q = i - r * (BK / 4);
      as[r * AS + q * 4 + 0] = pa[v].x;
      as[r * AS + q * 4 + 1] = pa[v].y;
      as[r * AS + q * 4 + 2] = pa[v].z;
      as[r * AS + q * 4 + 3] = pa[v].w;
    }
#pragma unroll
    for (int v = 0; v < NB4; ++v) {
      const int i = tid + v * 256;
      const int n = i / (BK / 4), q = i - n * (BK / 4);
      // B[k][n] = W1s[n][k0 + k]  (n = hidden index)
      bs[(q * 4 + 0) * H + n] = pb[v].x * a1;
      bs[(q * 4 + 1) * H + n] = pb[v].y * a1;
      bs[(q * 4 + 2) * H + n] = pb[v].z * a1;
      bs[(q * 4 + 3) * H + n] = pb[v].w * a1;
    }
  };
  stage_load(0);
  stage_store(0);
  __syncthreads();

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x16){0};

  for (int ch = 0; ch < nchunks; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < nchunks) stage_load((ch + 1) * BK);
    const float* __restrict__ as = as0 + buf * kMlpRows * AS + (wv * 32 + l31) * AS + half;
    const float* __restrict__ bs = bs0 + buf * BK * H + half * H + l31;
    // fragments for all BK/2 k-pairs of the chunk are read up front (registers), then the MFMAs run back to back
    float av[BK / 2], bv[BK / 2][NT];
#pragma unroll
    for (int kp = 0; kp < BK / 2; ++kp) {
      av[kp] = as[2 * kp];
#pragma unroll
      for (int t = 0; t < NT; ++t) bv[kp][t] = bs[(2 * kp) * H + t * 32];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kp = 0; kp < BK / 2; ++kp) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kp], bv[kp][t], acc[t], 0, 0, 0);
    }
    if (ch + 1 < nchunks) stage_store(buf ^ 1);
    lds_barrier();
  }

  // epilogue (re-using the staging buffers): g_h -> LDS, g_pre = g_h * silu'(pre) in place, then the NB-wide GEMV
  float* __restrict__ gp = smem;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = t * 32 + l31;  // hidden index
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int lr = wv * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;  // row within the block
      gp[lr * GS + col] = acc[t][r];
    }
  }
  __syncthreads();
  for (int i = tid; i < kMlpRows * H; i += 256) {
    const int r = i / H, k = i - r * H;
    float pre = 0.f;
    for (int c = 0; c < nb; ++c) pre += es[r * kMaxNb + c] * w0t[c * H + k];
    gp[r * GS + k] *= silu_grad_f(pre);
  }
  __syncthreads();
  for (int o = tid; o < kMlpRows * nb; o += 256) {
    const int r = o / nb, c = o - r * nb;
    if (blk0 + r >= E) continue;
    float s = 0.f;
#pragma unroll 8
    for (int k = 0; k < H; ++k) s += gp[r * GS + k] * w0s[k * kMaxNb + c];
    g_emb[(blk0 + r) * nb + c] = s;
  }
}

static int check_args(const void* emb, const void* W0, const void* W1, int nb, int H, int W, int64_t E,
                      const char* fn) {
  if (E < 0 || nb <= 0 || nb > kMaxNb || W <= 0 || (E > 0 && (!emb || !W0 || !W1))) {
    set_error(std::string(fn) + ": invalid argument");
    return NQA_ERR_INVALID;
  }
  if (H != 64 && H != 128) {
    set_error(std::string(fn) + ": hidden width must be 64 or 128 for the fused MFMA kernel");
    return NQA_ERR_UNSUPPORTED;
  }
  return NQA_OK;
}

}  // namespace nqa

using namespace nqa;

extern "C" {

int nqa_radial_mlp_supported(int32_t dtype, int32_t num_basis, int32_t hidden, int32_t out_features) {
  return (dtype == NQA_F32 && num_basis > 0 && num_basis <= kMaxNb && (hidden == 64 || hidden == 128) &&
          out_features > 0 && out_features % 4 == 0)
             ? 1
             : 0;
}

int nqa_radial_mlp_fwd(int32_t dtype, const void* edge_embedding, const void* w0, double alpha0, const void* w1,
                       double alpha1, int32_t num_basis, int32_t hidden, int32_t out_features, int64_t num_edges,
                       void* edge_weight, nqa_stream stream) {
  if (dtype != NQA_F32) {
    set_error("nqa_radial_mlp_fwd: only float32 is implemented on MFMA");
    return NQA_ERR_UNSUPPORTED;
  }
  int rc = check_args(edge_embedding, w0, w1, num_basis, hidden, out_features, num_edges, "nqa_radial_mlp_fwd");
  if (rc != NQA_OK) return rc;
  if (num_edges == 0) return NQA_OK;
  if (edge_weight == nullptr || out_features % 4 != 0) {
    set_error("nqa_radial_mlp_fwd: invalid output (needs out_features % 4 == 0)");
    return NQA_ERR_INVALID;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  const unsigned grid = (unsigned)((num_edges + kMlpRows - 1) / kMlpRows);
  const float* e = static_cast<const float*>(edge_embedding);
  const float* a = static_cast<const float*>(w0);
  const float* b = static_cast<const float*>(w1);
  float* o = static_cast<float*>(edge_weight);
  if (hidden == 128)
    hipLaunchKernelGGL(radial_mlp_fwd_kernel<128>, dim3(grid), dim3(256), 0, s, e, a, b, (float)alpha0,
                       (float)alpha1, num_basis, out_features, num_edges, o);
  else
    hipLaunchKernelGGL(radial_mlp_fwd_kernel<64>, dim3(grid), dim3(256), 0, s, e, a, b, (float)alpha0,
                       (float)alpha1, num_basis, out_features, num_edges, o);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_error(std::string("nqa_radial_mlp_fwd: ") + hipGetErrorString(err));
    return NQA_ERR_LAUNCH;
  }
  return NQA_OK;
}

int nqa_radial_mlp_bwd(int32_t dtype, const void* edge_embedding, const void* w0, double alpha0, const void* w1,
                       double alpha1, const void* grad_edge_weight, int32_t num_basis, int32_t hidden,
                       int32_t out_features, int64_t num_edges, void* grad_edge_embedding, nqa_stream stream) {
  if (dtype != NQA_F32) {
    set_error("nqa_radial_mlp_bwd: only float32 is implemented on MFMA");
    return NQA_ERR_UNSUPPORTED;
  }
  int rc = check_args(edge_embedding, w0, w1, num_basis, hidden, out_features, num_edges, "nqa_radial_mlp_bwd");
  if (rc != NQA_OK) return rc;
  if (num_edges == 0) return NQA_OK;
  if (grad_edge_weight == nullptr || grad_edge_embedding == nullptr || out_features % 4 != 0) {
    set_error("nqa_radial_mlp_bwd: invalid argument (needs out_features % 4 == 0)");
    return NQA_ERR_INVALID;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  const unsigned grid = (unsigned)((num_edges + kMlpRows - 1) / kMlpRows);
  const float* e = static_cast<const float*>(edge_embedding);
  const float* a = static_cast<const float*>(w0);
  const float* b = static_cast<const float*>(w1);
  const float* g = static_cast<const float*>(grad_edge_weight);
  float* o = static_cast<float*>(grad_edge_embedding);
  if (hidden == 128)
    hipLaunchKernelGGL(radial_mlp_bwd_kernel<128>, dim3(grid), dim3(256), 0, s, e, a, b, g, (float)alpha0,
                       (float)alpha1, num_basis, out_features, num_edges, o);
  else
    hipLaunchKernelGGL(radial_mlp_bwd_kernel<64>, dim3(grid), dim3(256), 0, s, e, a, b, g, (float)alpha0,
                       (float)alpha1, num_basis, out_features, num_edges, o);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_error(std::string("nqa_radial_mlp_bwd: ") + hipGetErrorString(err));
    return NQA_ERR_LAUNCH;
  }
  return NQA_OK;
}

}  // extern "C"
