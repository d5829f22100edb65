/*
 * nequip_amd.h -- C ABI of libnequip_amd.so, the MI355X (gfx950) implementation of the NequIP
 * message-passing hot path.
 *
 * The reference (mir-group/nequip v0.19.0) is pure Python and has no FFI; its seam for accelerated
 * kernels is the torch.nn.Module contract of `TensorProductScatter` plus the `enable_*` model
 * modifiers (nequip/nn/_tp_scatter_base.py:9-109, adapters nequip/nn/_tp_scatter_oeq.py:4-57 and
 * nequip/nn/_tp_scatter_cueq.py:66-122).  This header is the native boundary that sits *under* that
 * contract: every entry point below names the reference interface it replaces.  The Python host side
 * (nequip_amd/nn/...) binds these symbols with ctypes and mirrors the reference module API on top.
 *
 * Conventions
 *  - Every data pointer is a *borrowed device pointer* (HIP global memory) owned by the caller; the
 *    library never allocates or frees result buffers.  Scratch comes from caller-provided workspaces
 *    sized by the matching *_workspace_bytes query.
 *  - Kernels are enqueued asynchronously on the passed `stream` (a hipStream_t); no entry point
 *    synchronises the device and there is no global mutable state besides immutable plans.
 *  - Return value: NQA_OK (0) or a negative error code; nqa_last_error() returns a thread-local
 *    human-readable message.  The host side converts non-zero codes to RuntimeError, matching the
 *    reference's exception-only error convention.
 *  - Feature layout is e3nn "mul_ir" ([mul, 2l+1], m fastest) unless the plan was created with
 *    NQA_LAYOUT_IR_MUL for that operand.  Edge attributes have mul == 1 per irrep.
 *  - dtype selects the arithmetic/storage type of feature tensors: NQA_F32 or NQA_F64 (the reference's
 *    `model_dtype`, nequip/nn/_tp_scatter_base.py:33).  Edge vectors are always float64
 *    (nequip/utils/global_dtype.py:5).
 */
#ifndef NEQUIP_AMD_H
#define NEQUIP_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NQA_ABI_VERSION 1

enum nqa_status {
  NQA_OK = 0,
  NQA_ERR_INVALID = -1,     /* bad argument / inconsistent irreps or instruction            */
  NQA_ERR_UNSUPPORTED = -2, /* l > NQA_LMAX, unsupported dtype, ...                          */
  NQA_ERR_LAUNCH = -3,      /* HIP launch / runtime failure                                  */
  NQA_ERR_WORKSPACE = -4    /* workspace missing or too small                                */
};

enum nqa_dtype { NQA_F32 = 0, NQA_F64 = 1 };
enum nqa_layout { NQA_LAYOUT_MUL_IR = 0, NQA_LAYOUT_IR_MUL = 1 };

enum nqa_plan_field {
  NQA_PLAN_DIM_IN1 = 0,      /* feature_irreps_in.dim                                        */
  NQA_PLAN_DIM_IN2 = 1,      /* irreps_edge_attr.dim                                         */
  NQA_PLAN_DIM_OUT = 2,      /* irreps_mid.dim                                               */
  NQA_PLAN_WEIGHT_NUMEL = 3, /* e3nn TensorProduct.weight_numel                              */
  NQA_PLAN_NUM_INSTR = 4,
  NQA_PLAN_OUT_NEEDS_ZERO = 5, /* 1 if output slots are shared/uncovered (caller must zero `out`) */
  NQA_PLAN_YPART_WIDTH = 6,  /* columns of the per-edge dY partial buffer (bwd_edge workspace) */
  NQA_PLAN_HAS_SPECIALIZED = 7, /* 1 if structure-specialised (edge-outer) kernels are prebuilt for this plan */
  NQA_PLAN_FUSED_ROWS_OK = 8  /* 1 if nqa_tp_scatter_bwd_fused keeps its per-channel operands (grad_out row, two feature
                                 rows, the weights and their gradient) in registers; 0: it spills (l_max = 4 middle
                                 layer: 364 values) and nqa_tp_scatter_bwd_x + _bwd_edge are the faster route */
};

typedef struct nqa_plan nqa_plan;
typedef void* nqa_stream; /* hipStream_t */

int nqa_abi_version(void);
const char* nqa_last_error(void);
/* largest supported l for features / edge attributes / outputs, and for spherical harmonics */
int nqa_lmax(void);
int nqa_sh_lmax(void);

/* ---------------------------------------------------------------------------------------------
 * Plan: replaces the constructor of nequip.nn.TensorProductScatter
 *   __init__(feature_irreps_in, irreps_edge_attr, irreps_mid, instructions)
 *   (nequip/nn/_tp_scatter_base.py:10-33), i.e. e3nn TensorProduct(..., 'uvu' instructions,
 *   shared_weights=False, internal_weights=False, irrep_normalization="component",
 *   path_normalization="element").  Built once from exactly those four arguments; immutable and
 *   thread-safe afterwards.  Each irreps operand is given as parallel arrays (mul, l, p) with
 *   p = +1 (even) / -1 (odd).  Instructions are (i_in1, i_in2, i_out) index triples, mode "uvu";
 *   path_weight may be NULL (all 1.0).  The weight tensor is consumed in instruction-list order,
 *   mul_in1 values per instruction.
 * ------------------------------------------------------------------------------------------- */
int nqa_plan_create(int32_t n_in1, const int32_t* in1_mul, const int32_t* in1_l, const int32_t* in1_p,
                    int32_t n_in2, const int32_t* in2_mul, const int32_t* in2_l, const int32_t* in2_p,
                    int32_t n_out, const int32_t* out_mul, const int32_t* out_l, const int32_t* out_p,
                    int32_t n_instr, const int32_t* instr_i1, const int32_t* instr_i2,
                    const int32_t* instr_io, const double* instr_path_weight,
                    int32_t layout_in1, int32_t layout_out, nqa_plan** plan);
void nqa_plan_destroy(nqa_plan* plan);
int64_t nqa_plan_query(const nqa_plan* plan, int32_t field);
/* The kernels read the plan's path tables from device memory.  The library does not allocate device
 * memory: the caller obtains the (position independent) table image, copies it into a device buffer it
 * owns (the host side keeps it as a non-persistent module buffer so it follows `.to(device)`) and
 * passes that buffer as `plan_image` to the nqa_tp_* calls. */
int64_t nqa_plan_image_bytes(const nqa_plan* plan);
int nqa_plan_image_write(const nqa_plan* plan, void* host_dst, int64_t host_dst_bytes);

/* ---------------------------------------------------------------------------------------------
 * Edge topology (CSR): replaces the implicit index handling of
 *   x[edge_src] (row gather, nequip/nn/_tp_scatter_base.py:36) and
 *   scatter(edge_features, edge_dst, dim=0, dim_size=N) (nequip/nn/utils.py:24-53),
 * for arbitrary (unsorted, repeated) int64 indices as in
 *   tests/unit/nn/test_tp_scatter_kernel.py:144-149.  Groups the E edges by `key` with a stable
 * radix sort:  rowptr[n]..rowptr[n+1] indexes the edges whose key == n (ascending original edge
 * id), edge_id[] holds their original ids and other_sorted[] the matching `other` index.
 * Call once with (key=edge_dst, other=edge_src) for forward / edge gradients and once with
 * (key=edge_src, other=edge_dst) for the feature gradient (the transposed graph, cf.
 * nequip/data/transforms/neighborlist.py:150-155 `edge_transpose_perm`).
 * Returns NQA_ERR_INVALID (reported asynchronously as out-of-range being clamped is NOT done):
 * indices must satisfy 0 <= idx < num_nodes; this is checked on the device and reported through
 * `status_flag` (int32 device word, set non-zero on violation) when it is non-NULL.
 * ------------------------------------------------------------------------------------------- */
int64_t nqa_csr_workspace_bytes(int64_t num_nodes, int64_t num_edges);
int nqa_csr_build(const int64_t* key, const int64_t* other, int64_t num_nodes, int64_t num_edges,
                  int32_t* rowptr, int32_t* edge_id, int32_t* other_sorted, int32_t* status_flag,
                  void* workspace, int64_t workspace_bytes, nqa_stream stream);

/* ---------------------------------------------------------------------------------------------
 * Fused gather -> Clebsch-Gordan tensor product ('uvu', per-edge weights) -> scatter-add.
 * Replaces TensorProductScatter.forward(x, edge_attr, edge_weight, edge_dst, edge_src)
 *   (nequip/nn/_tp_scatter_base.py:35-38):
 *   out[n, slot_p, u, k] = sum_{e: dst(e)=n} c_p * w[e, p, u] * sum_ij C^p_ijk x[src(e), u, i] y[e, j]
 * x [N, dim_in1], y [E, dim_in2], w [E, weight_numel], out [N, dim_out]; (rowptr, edge_id,
 * src_sorted) from nqa_csr_build(key=dst, other=src).  Rows of `out` without incoming edges are
 * written as zeros.  If NQA_PLAN_OUT_NEEDS_ZERO the caller must pre-zero `out`.
 * ------------------------------------------------------------------------------------------- */
int nqa_tp_scatter_fwd(const nqa_plan* plan, const void* plan_image, int32_t dtype, const void* x, const void* y, const void* w,
                       const int32_t* rowptr_dst, const int32_t* edge_id_dst, const int32_t* src_sorted,
                       void* out, int64_t num_nodes, int64_t num_edges, nqa_stream stream);

/* Backward of the op above w.r.t. the per-edge operands (replaces the autograd of the e3nn einsum
 * chain exercised by tests/unit/nn/test_tp_scatter_kernel.py:160-177):
 *   gw[e,p,u] = c_p sum_ijk C^p_ijk x[src(e),u,i] y[e,j] g[dst(e),slot_p,u,k]
 *   gy[e,j]   = sum_p sum_u c_p w[e,p,u] sum_ik C^p_ijk x[src(e),u,i] g[dst(e),slot_p,u,k]
 * gw and/or gy may be NULL (gradient not needed).  `workspace` (>= nqa_tp_bwd_edge_workspace_bytes)
 * holds deterministic per-path partial sums of gy; required iff gy != NULL. */
int64_t nqa_tp_bwd_edge_workspace_bytes(const nqa_plan* plan, int32_t dtype, int64_t num_edges);
int nqa_tp_scatter_bwd_edge(const nqa_plan* plan, const void* plan_image, int32_t dtype, const void* x, const void* y, const void* w,
                            const void* grad_out, const int32_t* rowptr_dst, const int32_t* edge_id_dst,
                            const int32_t* src_sorted, void* grad_w, void* grad_y, void* workspace,
                            int64_t workspace_bytes, int64_t num_nodes, int64_t num_edges, nqa_stream stream);

/* Backward w.r.t. the node features (a scatter over *src*: the transposed graph):
 *   gx[m, u, i] = sum_{e: src(e)=m} sum_p c_p w[e,p,u] sum_jk C^p_ijk y[e,j] g[dst(e),slot_p,u,k]
 * (rowptr, edge_id, dst_sorted) from nqa_csr_build(key=src, other=dst). */
int nqa_tp_scatter_bwd_x(const nqa_plan* plan, const void* plan_image, int32_t dtype, const void* y, const void* w, const void* grad_out,
                         const int32_t* rowptr_src, const int32_t* edge_id_src, const int32_t* dst_sorted,
                         void* grad_x, int64_t num_nodes, int64_t num_edges, nqa_stream stream);

/* All three gradients of the op in one pass over grad_out (same maths as the two calls above, i.e. the
 * autograd of nequip/nn/_tp_scatter_base.py:77-120 w.r.t. x, edge_attr and the weights): the edge kernel
 * (dst-CSR, grad_out[dst] resident in registers) additionally emits each edge's contribution to
 * grad_x[src(e)] as a row of `workspace`; a second kernel sums those rows per source node through the
 * src-CSR (rowptr_src, edge_id_src) -- deterministic, no atomics, and grad_out rows are never gathered
 * per edge.  grad_w, grad_y and grad_x are all required (callers needing fewer use the calls above).  Only plans with a structure-specialised
 * kernel (NQA_PLAN_HAS_SPECIALIZED) in float32 support it: nqa_tp_bwd_fused_workspace_bytes returns -1
 * otherwise and the caller uses nqa_tp_scatter_bwd_edge + nqa_tp_scatter_bwd_x. */
int64_t nqa_tp_bwd_fused_workspace_bytes(const nqa_plan* plan, int32_t dtype, int64_t num_edges);
int nqa_tp_scatter_bwd_fused(const nqa_plan* plan, const void* plan_image, int32_t dtype, const void* x, const void* y,
                             const void* w, const void* grad_out, const int32_t* rowptr_dst,
                             const int32_t* edge_id_dst, const int32_t* src_sorted, const int32_t* rowptr_src,
                             const int32_t* edge_id_src, void* grad_w, void* grad_y, void* grad_x, void* workspace,
                             int64_t workspace_bytes, int64_t num_nodes, int64_t num_edges, nqa_stream stream);

/* ---------------------------------------------------------------------------------------------
 * Edge vectors: replaces with_edge_vectors_ (nequip/nn/utils.py:68-118),
 *   edge_vec[e] = pos[edge_src[e]] - pos[edge_dst[e]] (+ edge_cell_shift[e] @ cell[batch[edge_dst[e]]]),
 * (edge_dst = edge_index[0] = centre atom, edge_src = edge_index[1] = neighbour), all float64.
 * cell [F,3,3] (rows = lattice vectors), edge_cell_shift [E,3] and batch [N] are optional (NULL: no
 * periodic images / single frame).  The adjoint nqa_edge_vectors_bwd replaces the autograd of those
 * index_select / baddbmm ops (the float64 atomic index_add kernels) by ordered per-atom sums over the
 * two CSRs of nqa_csr_build: g_pos [N,3], and -- when g_cell_per_node [N,9] is non-NULL -- the per-atom
 * contribution sum_{e: dst(e)=n} shift[e]^T g[e] to d/dcell (summed per frame by the caller).
 * ------------------------------------------------------------------------------------------- */
int nqa_edge_vectors_fwd(const double* pos, const int64_t* edge_dst, const int64_t* edge_src,
                         const double* edge_cell_shift, const double* cell, const int64_t* batch,
                         int64_t num_edges, double* edge_vec, nqa_stream stream);
int nqa_edge_vectors_bwd(const double* g_edge_vec, const double* edge_cell_shift, const int32_t* rowptr_dst,
                         const int32_t* edge_id_dst, const int32_t* rowptr_src, const int32_t* edge_id_src,
                         int64_t num_nodes, double sign, double* g_pos, double* g_cell_per_node, nqa_stream stream);
/* `sign` multiplies g_pos (-1: forces = -dE/dpos directly).  g_cell_per_node[n] = sum_{e: centre(e)=n} left[e] (x) g[e]
 * with left = edge_cell_shift -- or any other [E,3] per-edge factor: with left = edge_vec it is the per-atom virial
 * sum, which nqa_virial_finalize reduces per frame:
 *   virial[f] = -sym(sum_n per_atom[n]),  stress[f] = sym(...) / |det cell[f]|   (nequip/nn/grad_output.py:222-271;
 *   stress may be NULL (no cell); batch [N] int64 is required iff num_frames > 1). */
int nqa_virial_finalize(const double* per_atom, const int64_t* batch, const double* cell, int64_t num_nodes,
                        int64_t num_frames, double* virial, double* stress, nqa_stream stream);
/* nqa_frame_sum: out[f, :] = sum_{n: batch[n] = f} rows[n, :]  (float64, width <= 16, every frame written; batch may be
 * NULL for a single frame).  The per-frame reductions of the training path -- autograd's backward of
 * `symmetric_displacement[batch]` and the d/dcell sum of with_edge_vectors_ (nequip/nn/grad_output.py:230-247,
 * nequip/nn/utils.py:88-114), which ATen evaluates as float64 atomics -- as one ordered tree reduction per frame. */
int nqa_frame_sum(const double* rows, const int64_t* batch, int64_t num_nodes, int32_t width, int64_t num_frames,
                  double* out, nqa_stream stream);

/* ---------------------------------------------------------------------------------------------
 * Edge embedding: real spherical harmonics + Bessel radial basis with polynomial cutoff.
 * Replaces, for precomputed float64 edge vectors (nequip/nn/utils.py:68-118),
 *   SphericalHarmonicEdgeAttrs.forward  (nequip/nn/embedding/_edge.py:193-198; e3nn
 *       SphericalHarmonics(irreps 0..lmax, normalize=True, normalization="component")),
 *   EdgeLengthNormalizer.forward        (nequip/nn/embedding/_edge.py:65-80),
 *   BesselEdgeLengthEncoding.forward    (nequip/nn/embedding/_edge.py:136-150),
 *   PolynomialCutoff.forward            (nequip/nn/embedding/cutoffs.py:17-27),
 *   ApplyFactor (2*pi/r_max^2)          (nequip/model/nequip_models.py:318-322).
 * All arithmetic is float64; results are rounded to `dtype` exactly where the reference casts
 * (.to(model_dtype)), then bessel*cutoff*factor is formed in `dtype`.
 *   sh  [E, (lmax+1)^2]   emb [E, num_bessels]   cutoff [E]   (any output may be NULL)
 * rmax_recip_edge (optional, [E] float64) overrides the scalar 1/r_max per edge (per-edge-type
 * cutoffs).  bessel_weights: [num_bessels] float64 DEVICE pointer (the reference's buffer /
 * trainable parameter, _edge.py:112-121).
 * ------------------------------------------------------------------------------------------- */
int nqa_edge_embed_fwd(int32_t dtype, int32_t lmax, const double* edge_vec, int64_t num_edges,
                       double rmax_recip, const double* rmax_recip_edge, int32_t num_bessels,
                       const double* bessel_weights, double cutoff_p, double factor, void* sh, void* emb,
                       void* cutoff, nqa_stream stream);
/* Vector-Jacobian product of the above: g_edge_vec[E,3] (float64) from g_sh / g_emb (either may be
 * NULL).  This is the edge -> position leg of the force backward (nequip/nn/grad_output.py:217-221). */
int nqa_edge_embed_bwd(int32_t dtype, int32_t lmax, const double* edge_vec, int64_t num_edges,
                       double rmax_recip, const double* rmax_recip_edge, int32_t num_bessels,
                       const double* bessel_weights, double cutoff_p, double factor, const void* g_sh,
                       const void* g_emb, double* g_edge_vec, nqa_stream stream);
/* The same two kernels for an edge list with a reverse-edge pairing (nqa_edge_pairs: pair_row[e] < num_pairs for the
 * representative edge of a pair, its row in the per-pair arrays; >= num_pairs for the reverse edge): the radial rows are
 * ALSO written per pair, emb_pairs [P, num_bessels] = emb[rep_edge[p]] (what the radial MLP consumes when it is evaluated
 * once per pair -- the gather nqa_pair_gather would do), and the backward takes the per-pair cotangent g_emb_pairs [P,
 * num_bessels] directly (the adjoint nqa_pair_expand would do) next to / instead of the per-edge one; emb, g_sh, g_emb,
 * g_emb_pairs may be NULL.  Plain r_max only (no per-edge cutoffs: those lists are not paired).  Replaces, for a paired
 * list, `edge_embedding[rep]` / its autograd adjoint of nequip_amd's own host (nn/_paired_radial.py); the reference
 * evaluates edge_mlp per directed edge (nequip/nn/interaction_block.py:190-199). */
int nqa_edge_embed_fwd_paired(int32_t dtype, int32_t lmax, const double* edge_vec, int64_t num_edges,
                              double rmax_recip, int32_t num_bessels, const double* bessel_weights, double cutoff_p,
                              double factor, const int32_t* pair_row, int64_t num_pairs, void* sh, void* emb,
                              void* emb_pairs, nqa_stream stream);
int nqa_edge_embed_bwd_paired(int32_t dtype, int32_t lmax, const double* edge_vec, int64_t num_edges,
                              double rmax_recip, int32_t num_bessels, const double* bessel_weights, double cutoff_p,
                              double factor, const int32_t* pair_row, int64_t num_pairs, const void* g_sh,
                              const void* g_emb, const void* g_emb_pairs, double* g_edge_vec, nqa_stream stream);
/* Second order (force-matching training, nequip/nn/grad_output.py:220 create_graph=True): for the cotangent
 * cot_g_edge_vec [E,3] of the VJP output above, gg_sh / gg_emb = J(v) c (gradients w.r.t. g_sh / g_emb) and
 * g_edge_vec2 = (sum_k g_k Hessian_k(v)) c (gradient w.r.t. the edge vector).  Evaluated with forward-mode dual
 * numbers through the same per-edge code as the first-order kernels.  Any output may be NULL. */
int nqa_edge_embed_bwd_bwd(int32_t dtype, int32_t lmax, const double* edge_vec, int64_t num_edges,
                           double rmax_recip, const double* rmax_recip_edge, int32_t num_bessels,
                           const double* bessel_weights, double cutoff_p, double factor, const void* g_sh,
                           const void* g_emb, const double* cot_g_edge_vec, void* gg_sh, void* gg_emb,
                           double* g_edge_vec2, nqa_stream stream);

/* ---------------------------------------------------------------------------------------------
 * Radial network on MFMA: replaces ScalarMLPFunction.forward as InteractionBlock.edge_mlp
 *   (nequip/nn/interaction_block.py:119-127,196; nequip/nn/mlp.py:141-156,194-196,262-268) for the
 *   standard one-hidden-layer, bias-free, SiLU radial MLP:
 *     edge_weight[E, W] = silu(edge_embedding[E, nb] @ (w0[nb, H] * alpha0)) @ (w1[H, W] * alpha1)
 *   The hidden layer never leaves the chip.  `mode` selects how the big GEMM ([E,H] x [H,W]) runs on the matrix
 *   cores: NQA_MLP_FP32 = v_mfma_f32_32x32x2_f32 (bitwise an fp32 fma chain); NQA_MLP_BF16X6 = every fp32 operand
 *   split exactly into three bf16 terms and the six partial products >= 2^-16 accumulated in fp32 on
 *   v_mfma_f32_32x32x16_bf16 (dropped terms < 2^-24 relative: fp32 accuracy, 2.7x the fp32-MFMA ceiling).  The
 *   small first layer (K = nb) and all element-wise maths are fp32 in both modes.
 * nqa_radial_mlp_bwd is its vector-Jacobian product w.r.t. the edge embedding (inference forces:
 *   nequip/nn/grad_output.py:217-221); pre-activations are recomputed, not stored.  Parameter gradients
 *   (training) are not produced by these entry points -- the host side keeps the reference's mm/SiLU
 *   formulation in training mode.
 * nqa_radial_mlp_supported returns 1 when (dtype, nb, H, W) can run on the fused kernels
 *   (float32, nb <= 8, H in {64, 128}, W % 4 == 0).  `workspace` (>= nqa_radial_mlp_workspace_bytes(mode,
 *   backward, H, W); 0 for the fp32 forward) receives the re-laid-out / split second-layer weights (a function of
 *   w1, alpha1, mode and direction only): a caller whose weights are constant may keep it and pass
 *   workspace_ready != 0 on later calls to skip the prepass kernel.
 * ------------------------------------------------------------------------------------------- */
#define NQA_MLP_FP32 0
#define NQA_MLP_BF16X6 1
/* every entry point except nqa_radial_mlp_fwd_tangent (which rejects it): operands multiplied by exact powers of two and split into two fp16 terms, three partial products on
 * v_mfma_f32_32x32x16_f16 -- fp32-level accuracy (2^-22 per operand) at half the matrix instructions of BF16X6.  Forward:
 * weights scaled per 32-column tile, hidden activations per row.  Backward: weights per 32-row K chunk, the streamed
 * gradient rows by a running per-row exponent that is lowered, together with the row's accumulators, when a chunk
 * outgrows it.  The workspace of this mode differs from BF16X6's (nqa_radial_mlp_workspace_bytes). */
#define NQA_MLP_F16X3 2
/* OR-ed into `mode` of nqa_radial_mlp_bwd (ignored elsewhere): a scheduling hint, no effect on results beyond the order of
 * fp32 roundings -- "nothing else will run on the device next to this launch".  The inference backward of a NARROW output
 * (W <= 256) may then take the persistent form with all weight fragments resident in LDS (radial_mlp_pipe.h; enabled by
 * NQA_MLP_BWD_SMALL=1, see radial_mlp.hip for the measurements); without the hint it keeps the many-short-workgroups form, which shares the chip better with concurrent streams (nequip_amd's host runs
 * the radial backward of all layers but the first next to the main chain). */
#define NQA_MLP_HINT_DEVICE_IS_IDLE 0x100
int nqa_radial_mlp_supported(int32_t dtype, int32_t num_basis, int32_t hidden, int32_t out_features);
int64_t nqa_radial_mlp_workspace_bytes(int32_t mode, int32_t backward, int32_t hidden, int32_t out_features);
int nqa_radial_mlp_fwd(int32_t dtype, int32_t mode, const void* edge_embedding, const void* w0, double alpha0,
                       const void* w1, double alpha1, int32_t num_basis, int32_t hidden, int32_t out_features,
                       int64_t num_edges, void* edge_weight, void* workspace, int64_t workspace_bytes,
                       int32_t workspace_ready, nqa_stream stream);
int nqa_radial_mlp_bwd(int32_t dtype, int32_t mode, const void* edge_embedding, const void* w0, double alpha0,
                       const void* w1, double alpha1, const void* grad_edge_weight, int32_t num_basis,
                       int32_t hidden, int32_t out_features, int64_t num_edges, void* grad_edge_embedding,
                       void* workspace, int64_t workspace_bytes, int32_t workspace_ready, nqa_stream stream);

/* Training variants of the fused radial MLP (force-matching training differentiates the force pass once more,
 *   nequip/nn/grad_output.py:216-221 `create_graph=self.training`; in the reference all of this is autograd through
 *   ScalarMLPFunction.forward, nequip/nn/mlp.py:194-196).  NQA_MLP_BF16X6 only.  With W_i = w_i alpha_i, P = emb W0,
 *   G_h = grad_edge_weight W1^T:
 * nqa_radial_mlp_bwd_train, cotangent == NULL (first order, with the pieces of the parameter gradients):
 *     grad_edge_embedding = (G_h silu'(P)) W0^T                       [E, num_basis]
 *     hidden_out          = silu(P)                                   [E, hidden]   (dW1 = alpha1 hidden_out^T grad_edge_weight)
 *     w0_partials[tile]   = emb_tile^T (G_h silu'(P))_tile            [tiles][num_basis][hidden], dW0 = alpha0 sum_tiles
 *   cotangent != NULL ([E, num_basis], a cotangent c of grad_edge_embedding; second order), Q = c W0:
 *     grad_edge_embedding = (Q G_h silu''(P)) W0^T                    = d<c, g_emb>/d emb
 *     hidden_out          = Q silu'(P)                                (d<c,g_emb>/dW1 = alpha1 hidden_out^T grad_edge_weight)
 *     w0_partials[tile]   = emb^T (Q G_h silu''(P)) + c^T (G_h silu'(P))   (d<c,g_emb>/dW0 = alpha0 sum_tiles)
 *   tiles = nqa_radial_mlp_train_tiles(num_edges) (128-edge workgroup tiles).
 * nqa_radial_mlp_fwd_tangent: out = (Q silu'(P)) W1 = d<c, g_emb>/d grad_edge_weight, the directional derivative of
 *   the MLP along `cotangent`; same workspace as nqa_radial_mlp_fwd. */
/* nqa_radial_mlp_bwd_paired: nqa_radial_mlp_bwd for an incoming gradient given as two row streams that are added on
 *   the fly (grad_edge_weight[row] + grad_edge_weight2[row]): the halves written by the two directed edges of a pair in
 *   nqa_tp_scatter_bwd_*_paired.  num_edges counts rows (pairs).  NQA_MLP_BF16X6 or NQA_MLP_F16X3. */
int nqa_radial_mlp_bwd_paired(int32_t dtype, int32_t mode, const void* edge_embedding, const void* w0, double alpha0,
                              const void* w1, double alpha1, const void* grad_edge_weight,
                              const void* grad_edge_weight2, int32_t num_basis, int32_t hidden, int32_t out_features,
                              int64_t num_edges, void* grad_edge_embedding, void* workspace, int64_t workspace_bytes,
                              int32_t workspace_ready, nqa_stream stream);
int64_t nqa_radial_mlp_train_tiles(int64_t num_edges);
int nqa_radial_mlp_bwd_train(int32_t dtype, int32_t mode, const void* edge_embedding, const void* cotangent,
                             const void* w0, double alpha0, const void* w1, double alpha1,
                             const void* grad_edge_weight, int32_t num_basis, int32_t hidden, int32_t out_features,
                             int64_t num_edges, void* grad_edge_embedding, void* hidden_out, void* w0_partials,
                             void* workspace, int64_t workspace_bytes, int32_t workspace_ready, nqa_stream stream);
int nqa_radial_mlp_fwd_tangent(int32_t dtype, int32_t mode, const void* edge_embedding, const void* cotangent,
                               const void* w0, double alpha0, const void* w1, double alpha1, int32_t num_basis,
                               int32_t hidden, int32_t out_features, int64_t num_edges, void* out, void* workspace,
                               int64_t workspace_bytes, int32_t workspace_ready, nqa_stream stream);

/* The last layer of a DEEPER radial MLP (ScalarMLPFunction with hidden_layers_depth >= 2, nequip/nn/mlp.py:81-196; the
 *   reference's tutorial model uses depth 2 / width 64, configs/tutorial.yaml:222-223) on the same GEMM cores:
 *     nqa_radial_mlp_last_fwd:  out [E, out_features] = silu(pre) @ (w alpha)
 *     nqa_radial_mlp_last_bwd:  grad_pre [E, hidden]  = ((grad_out (+ grad_out2)) @ (w alpha)^T) * silu'(pre)
 *   `pre` [E, hidden] are the PRE-activations of the layer's input -- for depth 2 the output of nqa_radial_mlp_fwd on the
 *   first two weight matrices (out_features = hidden width), whose gradient nqa_radial_mlp_bwd then takes on to the edge
 *   embedding; deeper stacks chain nqa_radial_mlp_last_fwd.  hidden 64 or 128, out_features % 4 == 0, float32, mode
 *   NQA_MLP_F16X3 (fp32-accurate two-plane fp16 split; workspaces as for nqa_radial_mlp_fwd / _bwd in that mode).
 *   Inference (first order) only: training-mode deep MLPs stay on the ATen formulation. */
int nqa_radial_mlp_last_fwd(int32_t dtype, int32_t mode, const void* pre, const void* w, double alpha, int32_t hidden,
                            int32_t out_features, int64_t num_edges, void* out, void* workspace, int64_t workspace_bytes,
                            int32_t workspace_ready, nqa_stream stream);
int nqa_radial_mlp_last_bwd(int32_t dtype, int32_t mode, const void* pre, const void* w, double alpha,
                            const void* grad_out, const void* grad_out2, int32_t hidden, int32_t out_features,
                            int64_t num_edges, void* grad_pre, void* workspace, int64_t workspace_bytes,
                            int32_t workspace_ready, nqa_stream stream);


/* ---------------------------------------------------------------------------------------------
 * Node-side channel mixing in one launch: replaces e3nn o3.Linear (linear_1 / linear_2,
 *   nequip/nn/interaction_block.py:82-87,129-138,177,201), the self-connection
 *   FullyConnectedTensorProduct with scalar node attributes (interaction_block.py:142-146,175; weights
 *   pre-contracted per atom type by the caller) and the residual add `x + sc` (:203-204):
 *     out[z, ob, w, m] = scale * sum_{(ib->ob)} sum_u x[z, ib, u, m] * W[type(z)][ib->ob][u, w]  (+ addend[z,...])
 *   mul_ir layout.  chunk_table: int32 records {o_off, d, mul_out, c0, instr_begin, instr_end, width, 0} (one per
 *   64-channel chunk of an output irrep block, every output element covered exactly once); instr_table: int32
 *   records {x_off, mul_in, w_off, 0} ([mul_in, mul_out] row-major matrix at weights + type*weight_stride + w_off).
 *   chunk_width = 64: float32 runs on fp32 MFMA with LDS-staged operands, float64 on the VALU kernel;
 *   chunk_width = -64 forces the VALU kernel for float32 as well.
 *   atom_types (int64 [N]) is required iff n_types > 1.  The backward w.r.t. x is the same call with transposed
 *   tables/weights.  chunk_table / instr_table are HOST pointers (<= 64 instructions; more than 40 chunks are
 *   processed in several launches): they are copied into the kernel arguments at call time.  x, weights, addend, out, atom_types are device pointers.
 * nqa_gate: e3nn Gate (nequip/nn/convnetlayer.py:104-112,162-164): in = scalars (+) gates (+) gated ->
 *   out = act(scalars) (+) act(gates)[u] * gated[u, :] (act 0 = identity, 1 = silu, 2 = tanh, each times its e3nn
 *   normalize2mom constant `cst`).  col_table: one 32-byte record {int32 a, b, c, d; double cst; int32 e, f} per
 *   column -- forward (backward == 0), per OUTPUT column: {src, gate (-1 for scalars), act, 0, cst, 0, 0}:
 *   out[z,c] = gate < 0 ? act(in[z,src]) : act(in[z,gate]) * in[z,src]; backward, per INPUT column:
 *   {kind, act, o, i, cst, len, gate}: kind 0 (scalar) gin = g[z,o] act'(in[z,c]); kind 1 (gate) gin = act'(in[z,c])
 *   sum_{m<len} g[z,o+m] in[z,i+m]; kind 2 (gated) gin = act(in[z,gate]) g[z,o]; kind 3: zero.
 *   Second order (training, nequip/nn/grad_output.py:220 create_graph): with a cotangent c [N, dim_in] of gin,
 *   backward = 2 writes d<c,gin>/d grad_out [N, dim_out] (forward table), backward = 3 writes d<c,gin>/d input
 *   [N, dim_in] (backward table).
 * ------------------------------------------------------------------------------------------- */
int nqa_node_linear(int32_t dtype, const void* x, const void* weights, const void* addend, void* out,
                    const int64_t* atom_types, const void* chunk_table, int32_t n_chunks, const void* instr_table,
                    int32_t n_instr, int32_t n_types, int64_t weight_stride, int32_t dim_in, int32_t dim_out,
                    int64_t num_nodes, double scale, int32_t chunk_width, nqa_stream stream);
/* ... _ordered: the same launch with an atom order (int32 [N] device pointer, may be NULL; float32 MFMA kernels only): the
 *   work units walk the atoms in that order and skip the typed stages of atom types none of their atoms has.  ANY permutation
 *   gives the same results; one that groups the atoms by type makes a typed map (the self-connection: one weight set per
 *   atom type) one pass per atom instead of one per type. */
int nqa_node_linear_ordered(int32_t dtype, const void* x, const void* weights, const void* addend, void* out,
                            const int64_t* atom_types, const int32_t* atom_order, const void* chunk_table, int32_t n_chunks,
                            const void* instr_table, int32_t n_instr, int32_t n_types, int64_t weight_stride,
                            int32_t dim_in, int32_t dim_out, int64_t num_nodes, double scale, int32_t chunk_width,
                            nqa_stream stream);
int nqa_gate(int32_t dtype, int32_t backward, const void* input, const void* grad_out, const void* cotangent,
             void* out, const void* col_table, int32_t dim_in, int32_t dim_out, int64_t num_nodes, nqa_stream stream);

/* ---------------------------------------------------------------------------------------------
 * Split 16-bit form of nqa_node_linear (float32 tensors, default for constant weights): the same maps -- e3nn `o3.Linear`
 *   / the type-contracted `FullyConnectedTensorProduct` of nequip/nn/interaction_block.py:82-87,129-146,175-177,201 --
 *   with fp32-level accuracy on the 16-bit matrix pipe.  Default: every operand multiplied by an exact power of two and
 *   written as the sum of two fp16 numbers, three partial products on v_mfma_f32_32x32x16_f16 (weights scaled per atom
 *   type, instruction and 16-row K block; every output column carries a running exponent).  NQA_NODE_F16=0 (environment,
 *   read at every call): three bf16 numbers per operand, six partial products on v_mfma_f32_32x32x16_bf16.
 * nqa_node_weights_pack: weights [n_types][weight_stride] (the layout nqa_node_linear reads) -> `packed`
 *   (nqa_node_weights_pack_bytes bytes): [type][instruction][16-row K block][32-column tile][plane][lane] of 16 bytes = 8
 *   16-bit values W[16 kb + 8 (lane >> 5) + e][32 tile + (lane & 31)], zero beyond the matrix; fp16 mode: 2 planes, then
 *   int32 exponents [type][instruction][K block]; bf16 mode: 3 planes.  Once per weight version, and the buffer must be
 *   consumed in the mode it was packed in.
 * nqa_node_linear_packed: out = scale * sum_instr x_block @ W (+ addend), tables as for nqa_node_linear (chunks must
 *   start at multiples of 64 channels); one 64-lane workgroup per (64-channel chunk, floor(32 / d) atoms) unit.
 * ------------------------------------------------------------------------------------------- */
int64_t nqa_node_weights_pack_bytes(const void* chunk_table, int32_t n_chunks, const void* instr_table, int32_t n_instr,
                                    int32_t n_types);
int nqa_node_weights_pack(const void* weights, const void* chunk_table, int32_t n_chunks, const void* instr_table,
                          int32_t n_instr, int32_t n_types, int64_t weight_stride, void* packed, nqa_stream stream);
int nqa_node_linear_packed(const void* x, const void* packed, const void* addend, void* out, const int64_t* atom_types,
                           const void* chunk_table, int32_t n_chunks, const void* instr_table, int32_t n_instr,
                           int32_t n_types, int32_t dim_in, int32_t dim_out, int64_t num_nodes, double scale,
                           nqa_stream stream);
int nqa_node_linear_packed_ordered(const void* x, const void* packed, const void* addend, void* out,
                                   const int64_t* atom_types, const int32_t* atom_order, const void* chunk_table,
                                   int32_t n_chunks, const void* instr_table, int32_t n_instr, int32_t n_types,
                                   int32_t dim_in, int32_t dim_out, int64_t num_nodes, double scale, nqa_stream stream);

/* ---------------------------------------------------------------------------------------------
 * The node side of a layer boundary in one launch per direction.  Between two tensor products the reference runs
 *   h = linear_2(y) + sc;  x = Gate(h)                                   (nequip/nn/convnetlayer.py:156-170)
 *   sc' = sc(x, node_attrs);  x1 = linear_1(x) / sqrt(avg_num_neighbors)   (nequip/nn/interaction_block.py:175-181)
 * -- Gate, linear_1 and the self-connection as three passes over [N, D] rows (and four more in the backward).
 * nqa_node_fused runs up to two `parts` (each what nqa_node_linear_packed takes: rows, packed fp16-split weights and the
 *   tables they were packed with, a destination) in ONE launch:
 *     - parts[k].in_gate != NULL: the part's rows x are PRE-gate rows h; the instruction x_off of its tables are offsets
 *       into Gate(h) and the gate is applied while the operand slab is staged (forward: both consumers of a gate in
 *       one launch, two destinations, Gate(h) never written);
 *     - parts[1].accumulate != 0: parts[1] has the same output chunks as parts[0] and is ADDED into the same tiles
 *       (linear_1^T(g_x1) + sc^T(g_sc); one scale, out / addend / dim_out of parts[0]);
 *     - out_gate != NULL: the tiles are gradients w.r.t. Gate(h) and leave through the gate's backward,
 *       out[N, gate_dim] = d/dh, with gate_h the pre-gate rows (chunk o_off are offsets into Gate(h)).
 *   nqa_gate_block: one block of a gate's OUTPUT: `mul` channels of dimension `d` at out_off <- values at val_off of the
 *   input row, gate scalars at gate_off (one per channel; -1: a scalar block, act applied to the values; act / cst as
 *   for nqa_gate).  Gated blocks need d >= 3; gated / activated blocks need offsets, multiplicities and row widths that
 *   are multiples of 4 (NQA_ERR_INVALID otherwise: the caller keeps the separate launches).  Float32, fp16-split packing
 *   (NQA_ERR_UNSUPPORTED under NQA_NODE_F16=0); at most 48 merged instructions; tables are HOST pointers.
 *   atom_order (optional, int32 [N], device): the order in which the work units walk the atoms -- ANY permutation gives the
 *   same results; a unit skips the typed stages of atom types none of its atoms has, so an order that groups the atoms by
 *   type makes the typed self-connection (at most 16 types) one pass per atom instead of one per type.
 * nqa_node_fused_plan: the merged tables of such a launch, for host-side tests -- chunk records of 12 int32 {o_off, d,
 *   mul_out, c0, instr_begin, instr_end, dst, epilogue (0 plain, 1 scalar block, 2 gated block of the output gate), ev_off,
 *   eg_off, act, cst (float bits)}, instruction records of 8 int32 {x_off, mul_in, frag_off, exp_off, gate_off (>= 0 gate
 *   scalars, -1 activate the values, -2 plain), act, cst (float bits), operand set}; returns (n_chunks << 16) | n_instr.
 * ------------------------------------------------------------------------------------------- */
typedef struct nqa_gate_block {
  int32_t out_off, d, mul, val_off, gate_off, act;
  double cst;
} nqa_gate_block;

typedef struct nqa_node_part {
  const void* x;           /* [N, dim_in] float32 rows */
  const void* packed;      /* nqa_node_weights_pack of (chunk_table, instr_table, n_types), fp16 split */
  const void* chunk_table; /* as for nqa_node_linear_packed (host) */
  const void* instr_table;
  int32_t n_chunks, n_instr, n_types, dim_in;
  void* out;               /* [N, dim_out]; ignored when accumulate != 0 */
  const void* addend;      /* optional [N, dim_out] */
  int32_t dim_out, accumulate;
  double scale;
  const nqa_gate_block* in_gate; /* optional (host) */
  int32_t n_in_gate, pad;
} nqa_node_part;

int nqa_node_fused(const nqa_node_part* parts, int32_t n_parts, const int64_t* atom_types, const int32_t* atom_order,
                   int64_t num_nodes, const nqa_gate_block* out_gate, int32_t n_out_gate, const void* gate_h,
                   int32_t gate_dim, nqa_stream stream);
int nqa_node_fused_plan(const nqa_node_part* parts, int32_t n_parts, const nqa_gate_block* out_gate, int32_t n_out_gate,
                        int32_t* chunks_out, int32_t chunks_cap, int32_t* instr_out, int32_t instr_cap);

/* ---------------------------------------------------------------------------------------------
 * Per-atom energy head in one launch per direction.  After the last convolution the reference runs Gate (scalars only) ->
 *   ScalarMLP readout of depth 0 (x @ W alpha, nequip/nn/mlp.py:262-268; built at nequip/model/nequip_models.py:371-381)
 *   -> PerTypeScaleShift (float64 addcmul, nequip/nn/atomwise.py:116-284), and autograd the same chain backwards.
 *   backward == 0:  out[z] (float64) = shift[type z] + scale[type z] * double( sum_c W[c] cst act(h[z, c]) )
 *   backward != 0:  out[z, c] (float32) = float(grad_e[z] * scale[type z]) * W[c] * cst act'(h[z, c])
 *   h [N, dim] float32 (dim a multiple of 4), readout_weight [dim] float32 with alpha folded in, scales / shifts float64 of
 *   length 0 (absent), 1 (shared) or one per type (atom_types required), act as for nqa_gate (0 identity, 1 silu, 2 tanh).
 *   The dot product is accumulated in float32 and scaled / shifted in float64, as the reference does.
 * ------------------------------------------------------------------------------------------- */
int nqa_energy_head(int32_t backward, const void* h, const void* readout_weight, const void* scales, int32_t n_scales,
                    const void* shifts, int32_t n_shifts, const int64_t* atom_types, const void* grad_e, void* out,
                    int32_t dim, int32_t act, double cst, int64_t num_nodes, nqa_stream stream);

/* ---------------------------------------------------------------------------------------------
 * Paired radial weights.  InteractionBlock.edge_mlp (nequip/nn/interaction_block.py:119-127,190-192) is a function of
 *   the edge length alone, and a neighbour list holds every interaction as (i <- j, S) and (j <- i, -S): the reference
 *   evaluates the MLP twice per pair.  nqa_edge_pairs finds the pairs of a list:
 *     weight_rows[e] = p (representative edge of pair p: dst < src, or a self image with a "positive" shift) or
 *                      p + P (its reverse), P = num_edges / 2;  rep_edge[p] = the representative edge;
 *     partner_edge[e] = the reverse edge of e;
 *     *ok (device int32) = 1 iff every edge has exactly one reverse partner (else the arrays are not to be used).
 *   edge_cell_shift: [E, 3] float32 / float64 (shift_dtype) integer-valued, or NULL.  (rowptr_dst, edge_id_dst, src_sorted):
 *   the dst-CSR of the same list (nqa_csr_build) -- the reverse of (i <- j, S) is looked up in row j, eight lanes per edge,
 *   no sort; pairs are numbered in the edge order of their representative.  For nqa_edge_pairs edge_id_dst may be NULL: the
 *   dst-CSR then lists the edges in edge order (edge_id[k] == k: a list grouped by centre atom, as the device neighbour list
 *   emits it), one dependent load less per look-up.
 * nqa_csr_from_pairs: the by-SOURCE CSR of a paired list without sorting -- row j holds the partners of the edges of row j
 *   of the dst-CSR: rowptr_src = rowptr_dst, edge_id_src[k] = partner_edge[edge_id_dst[k]], dst_sorted = src_sorted.
 * nqa_tp_scatter_{fwd,bwd_edge,bwd_x,bwd_fused}_paired: the tensor-product entry points above with
 *   w = [num_pairs, weight_numel] (one row per pair) and, for the edge backward, grad_w = [2 * num_pairs, weight_numel]
 *   (row weight_rows[e] receives edge e's gradient; the caller -- or nqa_radial_mlp_bwd's second stream -- adds the two
 *   halves).  weight_rows is given in CSR slot order of the respective topology (weight_rows[edge_id[slot]]).
 *   Structure-specialised float32 plans only (NQA_ERR_UNSUPPORTED otherwise).
 * ------------------------------------------------------------------------------------------- */
int64_t nqa_edge_pairs_workspace_bytes(int64_t num_edges);
int nqa_edge_pairs(const int64_t* edge_dst, const int64_t* edge_src, const void* edge_cell_shift, int32_t shift_dtype,
                   const int32_t* rowptr_dst, const int32_t* edge_id_dst, const int32_t* src_sorted, int64_t num_edges,
                   int64_t num_nodes, void* workspace, int64_t workspace_bytes, int32_t* weight_rows, int64_t* rep_edge,
                   int32_t* partner_edge, int32_t* ok, nqa_stream stream);
int nqa_csr_from_pairs(const int32_t* edge_id_dst, const int32_t* partner_edge, int64_t num_edges, int32_t* edge_id_src,
                       nqa_stream stream);
/* nqa_pair_gather: rows_out[p, :] = rows_in[rep_edge[p], :] (the per-pair rows of a per-edge float32 array, e.g. the
 *   edge embedding fed to the radial MLP); nqa_pair_expand is its adjoint: edge_rows[e, :] = pair_rows[weight_rows[e], :]
 *   for representative edges and 0 for the reverse ones (every row written).  width = 32-bit words per row. */
int nqa_pair_gather(const void* rows_in, const int64_t* rep_edge, int64_t num_pairs, int32_t width, void* rows_out,
                    nqa_stream stream);
/* nqa_pair_owner_lists: the pair lists of nqa_tp_scatter_bwd_pairs (below) from a pairing (weight_rows, rep_edge of
 *   nqa_edge_pairs with *ok == 1) and the dst-CSR of the same edge list (nqa_csr_build: rowptr, edge_id, src_sorted).  Every
 *   directed edge is the owner-side or the other-side edge of its pair -- owner of {i <- j, j <- i}: i when (i < j) xor
 *   (i + j odd), a self-image pair through its representative edge -- so both lists of a node are subsequences of its CSR
 *   row: two counting passes, two prefix sums, two fill passes, no sort, no synchronisation.  Slots keep the CSR order.
 *   Outputs (int32, device): owner_rowptr [N + 1], pair_other / pair_row / pair_edge_in / pair_edge_out [P] by slot,
 *   other_rowptr [N + 1], other_slot [P] (the slots grouped by their other node). */
int64_t nqa_pair_owner_workspace_bytes(int64_t num_edges, int64_t num_nodes);
int nqa_pair_owner_lists(const int32_t* weight_rows, const int64_t* rep_edge, const int32_t* rowptr_dst,
                         const int32_t* edge_id_dst, const int32_t* src_sorted, int64_t num_edges, int64_t num_nodes,
                         void* workspace, int64_t workspace_bytes, int32_t* owner_rowptr, int32_t* pair_other,
                         int32_t* pair_row, int32_t* pair_edge_in, int32_t* pair_edge_out, int32_t* other_rowptr,
                         int32_t* other_slot, nqa_stream stream);
/* nqa_pair_owner_lists_guard: for callers that cannot wait for nqa_edge_pairs' verdict (a hipGraph capture): when *ok == 0 on
 *   the device, both row pointers are zeroed -- the pair-centric kernels then walk empty lists instead of unwritten slots.
 *   The evaluation is void in that case; the caller reads `ok` afterwards. */
int nqa_pair_owner_lists_guard(const int32_t* ok, int64_t num_nodes, int32_t* owner_rowptr, int32_t* other_rowptr,
                               nqa_stream stream);
int nqa_pair_expand(const void* pair_rows, const int32_t* weight_rows, int64_t num_edges, int64_t num_pairs,
                    int32_t width, void* edge_rows, nqa_stream stream);
int nqa_tp_scatter_fwd_paired(const nqa_plan* plan, const void* plan_image, int32_t dtype, const void* x, const void* y,
                              const void* w, const int32_t* rowptr_dst, const int32_t* edge_id_dst,
                              const int32_t* src_sorted, void* out, int64_t num_nodes, int64_t num_edges,
                              const int32_t* weight_rows, int64_t num_pairs, nqa_stream stream);
int nqa_tp_scatter_bwd_edge_paired(const nqa_plan* plan, const void* plan_image, int32_t dtype, const void* x,
                                   const void* y, const void* w, const void* grad_out, const int32_t* rowptr_dst,
                                   const int32_t* edge_id_dst, const int32_t* src_sorted, void* grad_w, void* grad_y,
                                   void* workspace, int64_t workspace_bytes, int64_t num_nodes, int64_t num_edges,
                                   const int32_t* weight_rows, int64_t num_pairs, nqa_stream stream);
int nqa_tp_scatter_bwd_fused_paired(const nqa_plan* plan, const void* plan_image, int32_t dtype, const void* x,
                                    const void* y, const void* w, const void* grad_out, const int32_t* rowptr_dst,
                                    const int32_t* edge_id_dst, const int32_t* src_sorted, const int32_t* rowptr_src,
                                    const int32_t* edge_id_src, void* grad_w, void* grad_y, void* grad_x,
                                    void* workspace, int64_t workspace_bytes, int64_t num_nodes, int64_t num_edges,
                                    const int32_t* weight_rows, int64_t num_pairs, nqa_stream stream);
int nqa_tp_scatter_bwd_x_paired(const nqa_plan* plan, const void* plan_image, int32_t dtype, const void* y,
                                const void* w, const void* grad_out, const int32_t* rowptr_src,
                                const int32_t* edge_id_src, const int32_t* dst_sorted, void* grad_x, int64_t num_nodes,
                                int64_t num_edges, const int32_t* weight_rows, int64_t num_pairs, nqa_stream stream);

/* ---------------------------------------------------------------------------------------------
 * Pair-centric backward of the tensor-product scatter with paired radial weights (all three gradients in one pass;
 *   replaces, like nqa_tp_scatter_bwd_fused_paired, autograd's backward of `out = scatter(tp(x[src], y, w), dst)`,
 *   nequip/nn/_tp_scatter_base.py:33-38, for the case where `w` holds one row per reverse-edge pair).
 *   Every pair p = {other -> owner, owner -> other} has an owner node; the owner's wavefront evaluates both directed
 *   edges, so the weight row is read once, grad_w = [num_pairs, weight_numel] leaves ALREADY SUMMED over the two
 *   directed edges (no halves), and one per-pair row of grad_x contributions (instead of one per edge) goes through the
 *   workspace.  The pair lists (int32, device), P = num_edges / 2 slots grouped by owner:
 *     owner_rowptr [N+1]; per slot: pair_other (the other node), pair_row (row of w / grad_w), pair_edge_in (the edge
 *     other -> owner, i.e. dst = owner), pair_edge_out (owner -> other);  other_rowptr [N+1] / other_slot [P]: the slots
 *     grouped by their `other` node.  Any assignment of owners is valid; a balanced one halves every node's list.
 *   grad_y [E, dim_in2] per directed edge, grad_x [N, dim_in1] or NULL (grad_w and grad_y only: other_rowptr /
 *   other_slot are then not read).  float32 structure-specialised plans whose register
 *   budget allows it: nqa_tp_bwd_pairs_workspace_bytes returns -1 otherwise (use the per-edge entry points).
 *   Round 6: for multiples of 64 channels (and at least as many pairs as nodes) the other node's grad_x contributions are
 *   summed by floating-point atomics into a zeroed [N, dim_in1] accumulator inside the same workspace instead of one row per
 *   pair + a row sum (other_rowptr / other_slot are then not read): grad_x is equal within fp32 rounding but NOT bit-identical
 *   from call to call (sums in arrival order).  NQA_PAIR_GX_ATOMIC=0 in the environment keeps the fixed-order rows,
 *   NQA_PAIR_RING=0 the register / plain-loop kernels of rounds 3-5 (both read at every call).
 * ------------------------------------------------------------------------------------------- */
int64_t nqa_tp_bwd_pairs_workspace_bytes(const nqa_plan* plan, int32_t dtype, int64_t num_edges);
int nqa_tp_scatter_bwd_pairs(const nqa_plan* plan, const void* plan_image, int32_t dtype, const void* x, const void* y,
                             const void* w, const void* grad_out, const int32_t* owner_rowptr,
                             const int32_t* pair_other, const int32_t* pair_row, const int32_t* pair_edge_in,
                             const int32_t* pair_edge_out, const int32_t* other_rowptr, const int32_t* other_slot,
                             void* grad_w, void* grad_y, void* grad_x, void* workspace, int64_t workspace_bytes,
                             int64_t num_nodes, int64_t num_edges, nqa_stream stream);

/* Second-order companion (force-matching training differentiates the backward once more, nequip/nn/grad_output.py:217-221):
 *   with cotangents x_cot [N, dim_in1] of grad_x and y_cot [E, dim_in2] of grad_y,
 *     grad_w = Bw(x_cot, y, grad_out) + Bw(x, y_cot, grad_out)   [num_pairs, weight_numel], summed over each pair,
 *     grad_y = By(x_cot, w, grad_out) (+ By(x, w_cot, grad_out) when w_cot [num_pairs, weight_numel] is given)  [E, dim_in2]
 *   in ONE pair-centric pass (the intermediate sum_k C_ijk grad_out_k serves both products) instead of two calls of
 *   nqa_tp_scatter_bwd_pairs and an addition.  Plans with a single-wavefront pair kernel only
 *   (nqa_tp_bwd_pairs_dual_supported); workspace as nqa_tp_bwd_pairs_workspace_bytes. */
int32_t nqa_tp_bwd_pairs_dual_supported(const nqa_plan* plan, int32_t dtype);
int nqa_tp_scatter_bwd_pairs_dual(const nqa_plan* plan, const void* plan_image, int32_t dtype, const void* x,
                                  const void* x_cot, const void* y, const void* y_cot, const void* w, const void* w_cot,
                                  const void* grad_out, const int32_t* owner_rowptr, const int32_t* pair_other,
                                  const int32_t* pair_row, const int32_t* pair_edge_in, const int32_t* pair_edge_out,
                                  void* grad_w, void* grad_y, void* workspace, int64_t workspace_bytes,
                                  int64_t num_nodes, int64_t num_edges, nqa_stream stream);

/* Forward-mode companion of the same second-order backward: the gradient w.r.t. grad_out of <cotangents, backward outputs>,
 *   out = F(x_cot, y, w) + F(x, y_cot, w) + F(x, y, w_cot)      (F = nqa_tp_scatter_fwd; the op is trilinear)
 *   in one pass over the edges instead of three forward launches and two additions.  A NULL cotangent drops its term (at
 *   least one is required); w / w_cot hold one row per edge, or per pair when weight_rows (dst-CSR slot order, as
 *   nqa_tp_scatter_fwd_paired) is given.  Structure-specialised float32 plans (nqa_tp_fwd_jvp_supported). */
int32_t nqa_tp_fwd_jvp_supported(const nqa_plan* plan, int32_t dtype);
int nqa_tp_scatter_fwd_jvp(const nqa_plan* plan, const void* plan_image, int32_t dtype, const void* x, const void* y,
                           const void* w, const void* x_cot, const void* y_cot, const void* w_cot,
                           const int32_t* rowptr_dst, const int32_t* edge_id_dst, const int32_t* src_sorted, void* out,
                           int64_t num_nodes, int64_t num_edges, const int32_t* weight_rows, int64_t num_pairs,
                           nqa_stream stream);
/* ... and the gradient w.r.t. x:  grad_x = Bx(y_cot, w, grad_out) + Bx(y, w_cot, grad_out)  in one pass over the source
 *   CSR (operands as nqa_tp_scatter_bwd_x / _paired; plans for which nqa_tp_fwd_jvp_supported holds). */
int nqa_tp_scatter_bwd_x_dual(const nqa_plan* plan, const void* plan_image, int32_t dtype, const void* y, const void* w,
                              const void* y_cot, const void* w_cot, const void* grad_out, const int32_t* rowptr_src,
                              const int32_t* edge_id_src, const int32_t* dst_sorted, void* grad_x, int64_t num_nodes,
                              int64_t num_edges, const int32_t* weight_rows, int64_t num_pairs, nqa_stream stream);

/* ---------------------------------------------------------------------------------------------
 * Parameter gradients of the dense maps of the path (training; in the reference these come out of autograd as the
 *   weight-side `mm` / `einsum` backward of ScalarLinearLayer.forward (nequip/nn/mlp.py:262-268), e3nn o3.Linear
 *   (nequip/nn/interaction_block.py:82-87,129-138) and the self-connection FullyConnectedTensorProduct (:142-146)):
 *     dW[t][out_off + i*N + j] = sum_{z < num_rows, type(z) == t} sum_{m < d} A[z*lda + a_off + i*d + m] *
 *                                                                              B[z*ldb + b_off + j*d + m]
 *   for every record {int32 a_off, b_off, M, N, d, out_off} of instr_table (HOST pointer, <= 64 records; i < M,
 *   j < N).  ScalarMLP layer: one record {0, 0, in, out, 1, 0} over the E edge rows; o3.Linear / FCTP: one record per
 *   (input block -> output block) matrix with d = 2l+1 over the N atom rows (mul_ir layout), row_types / n_types > 1
 *   for the per-atom-type pre-contracted self-connection weights.
 *   The reduction over rows is split into `splits` ranges (nqa_wgrad_splits suggests a count that fills the device;
 *   each range is covered by the four wavefronts of one workgroup, which add their tiles in a fixed order on chip);
 *   partials is [splits][n_types][out_stride] floats, every element covered by a record is written (the rest is
 *   left untouched), and the caller sums over the first axis -- deterministic, no atomics.  float32 only (fp32 MFMA;
 *   outputs wider than 64 rows run as fp32-accurate split-bf16 products, NQA_WGRAD_EXACT_FP32=1 keeps them on fp32 MFMA).
 * ------------------------------------------------------------------------------------------- */
int32_t nqa_wgrad_splits(const void* instr_table, int32_t n_instr, int32_t n_types, int64_t num_rows);
int nqa_wgrad(int32_t dtype, const void* a_rows, const void* b_rows, const int64_t* row_types,
              const void* instr_table, int32_t n_instr, int64_t lda, int64_t ldb, int64_t num_rows, int32_t n_types,
              int64_t out_stride, int32_t splits, void* partials, nqa_stream stream);

/* ---------------------------------------------------------------------------------------------
 * Neighbour list on the device: replaces _compute_neighborlist_single_frame
 *   (nequip/data/_nl.py:63-165), i.e. the CPU library call `neighbour_list("ijS", pbc, cell, positions, cutoff)`:
 *   all ordered pairs (i, j, S) with |pos[j] - pos[i] + S @ cell| < r_max except i == j with S == 0;
 *   edge_index[0] = i (convolution centre), edge_index[1] = j, S = integer lattice shifts (0 along non-periodic
 *   directions).  pos [N,3] float64 (anywhere in space), cell [3,3] float64 rows = lattice vectors (NULL or all-zero:
 *   no cell, only valid without periodicity), pbc int32[3] (device; NULL = none).  Triclinic cells, cells thinner than
 *   the cutoff (several images of one atom) and mixed periodicity are supported.
 * Two calls because the number of edges is data dependent:
 *   nqa_neighbor_list_count builds the cell grid in `workspace` (>= nqa_neighbor_list_workspace_bytes(N)) and writes
 *     rowptr [N+1] int32 (device): rowptr[i]..rowptr[i+1] = the edge range of centre atom i, rowptr[N] = E;
 *   the caller reads rowptr[N], allocates edge_index [2,E] int64 and edge_cell_shift [E,3] float64, and calls
 *   nqa_neighbor_list_fill with the same workspace.  Edges come out grouped by centre atom in ascending order
 *   (dst-sorted: rowptr IS the dst-CSR row pointer of the tensor-product kernels) and in a deterministic order
 *   within each atom.
 * ------------------------------------------------------------------------------------------- */
int64_t nqa_neighbor_list_workspace_bytes(int64_t num_atoms);
int nqa_neighbor_list_count(const double* pos, const double* cell, const int32_t* pbc, double r_max, int64_t num_atoms,
                            void* workspace, int64_t workspace_bytes, int32_t* rowptr, nqa_stream stream);
int nqa_neighbor_list_fill(const void* workspace, const int32_t* rowptr, int64_t num_atoms, int64_t num_edges,
                           int64_t* edge_index, double* edge_cell_shift, nqa_stream stream);
/* Capacity-padded variant: no read-back of the edge count, so neighbour list -> pairing -> model can be captured in ONE
 *   hipGraph and replayed with new positions (molecular dynamics; the reference rebuilds its list on the host every step,
 *   nequip/integrations/ase.py:125-160 -> data/_nl.py:63-165).  After nqa_neighbor_list_count, writes exactly `edge_capacity`
 *   (even) edges: the E_real = rowptr[N] real ones, and behind the real edges of each atom its share of the E_cap - E_real
 *   padding edges -- self-image pairs (i <- i, +S), (i <- i, -S) with S = k >= k0 cells along the shortest periodic lattice
 *   vector, k0 |a| > r_max.  Padding edges lie outside the polynomial cutoff (nequip/nn/embedding/cutoffs.py:23-27: exactly
 *   zero there, with zero slope), so their radial embedding, their bias-free radial-MLP weights
 *   (nequip/nn/interaction_block.py:119-127) and every derivative w.r.t. their length vanish: energies and forces are those
 *   of the unpadded list.  A cell is required (edge vectors of self images come from the shift alone).
 *   rowptr_padded [N+1] int32 = the dst-CSR row pointer of the padded list (rowptr_padded[N] = edge_capacity);
 *   src_sorted int32 [edge_capacity] (may be NULL) = edge_index[1] as the int32 neighbour array of that CSR (whose edge ids
 *   are 0 .. edge_capacity - 1 in order).
 *   status int32[2] (device, may be NULL): status[0] = 1 when the list did not fit (E_real > capacity, or capacity - E_real
 *   odd) -- the output then holds padding edges ONLY (all indices valid, results void) and the caller repeats the step with a
 *   larger capacity; status[1] = E_real. */
int nqa_neighbor_list_fill_padded(const void* workspace, const int32_t* rowptr, int64_t num_atoms, int64_t edge_capacity,
                                  int32_t* rowptr_padded, int64_t* edge_index, double* edge_cell_shift, int32_t* src_sorted,
                                  int32_t* status, nqa_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* NEQUIP_AMD_H */
