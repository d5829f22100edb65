/*
 * nequip_amd_torch.h -- what libnequip_amd_torch.so provides (nequip_amd/csrc/torch_ops/nequip_amd_torch.cpp).
 *
 * The library registers the inference subset of the `nequip_amd::` dispatcher ops from C++ (TORCH_LIBRARY), so that an
 * AOTInductor package of the HIP-backed model (`nequip_amd.utils.aot.aot_export_model`, the reference's
 * `nequip-compile --mode aotinductor`, nequip/scripts/compile.py:248-344) runs in a process without a Python
 * interpreter.  It replaces, for such hosts, the reference's `import_custom_ops_libs` step
 * (nequip/utils/aoti_metadata.py:40-54: importing the Python modules named by the package's `nequip_custom_ops_libs`
 * entry): dlopen / link this library before constructing torch::inductor::AOTIModelPackageLoader.
 *
 * Ops (schemas identical to the Python registrations; CUDA dispatch key = HIP on ROCm builds; no CPU kernels):
 *   tp_scatter_fwd(Tensor x, Tensor edge_attr, Tensor edge_weight, Tensor edge_dst, Tensor edge_src, str plan) -> Tensor
 *   tp_scatter_bwd(Tensor grad_out, Tensor x, Tensor edge_attr, Tensor edge_weight, Tensor edge_dst, Tensor edge_src,
 *                  str plan, bool need_x, bool need_y, bool need_w) -> (Tensor, Tensor, Tensor)
 *   edge_vectors(Tensor pos, Tensor? cell, Tensor edge_index, Tensor? shift, Tensor? batch) -> Tensor
 *   edge_vectors_adj(Tensor g_vec, Tensor edge_index, Tensor? shift, Tensor? batch, SymInt num_nodes, SymInt num_frames,
 *                    bool need_cell) -> (Tensor, Tensor)
 *   edge_embed_fwd(Tensor edge_vec, Tensor bessel_weights, int lmax, bool want_sh, bool want_emb, int nb,
 *                  float rmax_recip, float p, float factor, bool f32) -> (Tensor, Tensor)
 *   edge_embed_bwd(Tensor edge_vec, Tensor bessel_weights, Tensor g_sh, Tensor g_emb, <same configuration>) -> Tensor
 *   radial_mlp_fwd(Tensor emb, Tensor w0, Tensor w1, float alpha0, float alpha1) -> Tensor
 *   radial_mlp_bwd(Tensor emb, Tensor w0, Tensor w1, Tensor g, float alpha0, float alpha1) -> Tensor
 *   node_linear(Tensor x, Tensor wp, Tensor? addend, Tensor? types, str key, float scale, bool transposed) -> Tensor
 *   gate(Tensor x, str key) -> Tensor
 *   gate_bwd(Tensor x, Tensor g, str key) -> Tensor
 * and the fused forms an exported energy / forces graph is made of since round 4 (one op per convolution edge side, per layer
 * boundary, for the readout and for the force / virial tail; nequip_amd/nn/_radial_tp_ops.py, o3/_node_ops.py,
 * nn/_energy_head.py, nn/_force_ops.py):
 *   radial_tp_fwd(Tensor emb, Tensor x, Tensor edge_attr, Tensor w0, Tensor w1, float alpha0, float alpha1,
 *                 Tensor edge_dst, Tensor edge_src, Tensor? edge_shift, str plan) -> (Tensor, Tensor)
 *       radial MLP + tensor-product scatter; pairs the edge list when every edge has one reverse partner (decided here,
 *       on the host, once per topology entry) and then evaluates the MLP once per pair; second result = the [E / 2, W]
 *       weight rows, flat, kept for the backward (a placeholder when the list does not pair up)
 *   radial_tp_bwd(Tensor grad_out, Tensor emb, Tensor x, Tensor edge_attr, Tensor w_rows, Tensor w0, Tensor w1,
 *                 float alpha0, float alpha1, Tensor edge_dst, Tensor edge_src, Tensor? edge_shift, str plan,
 *                 bool need_emb, bool need_x, bool need_y) -> (Tensor, Tensor, Tensor)
 *   node_stage_fwd(Tensor h, Tensor types, Tensor wp1, Tensor wps, str gate_key, str lin_key, str sc_key, float scale)
 *                  -> (Tensor, Tensor)        x1 = scale * linear_1(Gate(h)), sc = self_connection(Gate(h), types)
 *   node_stage_bwd(Tensor g_x1, Tensor g_sc, Tensor h, Tensor types, Tensor wp1, Tensor wps, str gate_key, str lin_key,
 *                  str sc_key, float scale) -> Tensor
 *   energy_head_fwd(Tensor h, Tensor w, Tensor? scales, Tensor? shifts, Tensor types, int act, float cst) -> Tensor
 *   energy_head_bwd(Tensor g_e, Tensor h, Tensor w, Tensor? scales, Tensor types, int act, float cst) -> Tensor
 *   force_virial(Tensor g_vec, Tensor edge_vec, Tensor edge_index, Tensor? batch, Tensor? cell, SymInt num_nodes,
 *                SymInt num_frames) -> (Tensor, Tensor, Tensor)      forces, virial, stress from dE/d(edge vectors)
 * Derived weight images (packed fp16-split fragments, transposed copies, the radial MLP's split second layer) are built
 * once per CONSTANT weight tensor: the cache is keyed on the identity of the tensor's storage, which an AOTInductor
 * package's constant buffers keep from call to call (NQA_OP_CONSTANT_CACHE=0 switches it off; a weight buffer rewritten
 * in place outside torch needs a new package or that switch).
 * `plan` / `key` are the canonical texts of the module constructor arguments (nequip_amd/nn/_tp_scatter_ops.py::plan_key,
 * nequip_amd/o3/_node_ops.py::linear_key / gate_key) that an exported graph carries as string constants.
 *
 * The C functions below expose the host tables derived from those texts (tests compare them byte for byte with the Python
 * host's; no GPU involved).
 */
#ifndef NEQUIP_AMD_TORCH_H
#define NEQUIP_AMD_TORCH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 1: the ops of this process are (or would be) the ones defined by this library's TORCH_LIBRARY block */
int nqa_torch_ops_registered_here(void);

/* Edge-topology (CSR) cache of the tp_scatter / edge_vectors_adj ops -- the reuse contract.
 *
 * The ops derive two CSRs (by destination, by source) from the int64 edge_index they are handed and share them between the
 * ops of an evaluation.  An entry is keyed on the identity of the index tensors' storage, torch's version counter, data
 * pointers, strides and sizes.  A host that refills ONE persistent device buffer outside torch (raw hipMemcpy, Kokkos
 * kernels: the usual LAMMPS-style set-up) changes none of these, so by DEFAULT (mode 1) an entry is reused only within the
 * evaluation that built it: `edge_embed_fwd`, which every exported energy graph runs once before any op that needs a CSR,
 * starts a new evaluation, as does nqa_torch_begin_evaluation().  Cost: two radix sorts per evaluation (0.24 ms at 400 k
 * edges).
 *   mode 0  never reuse (also NQA_TOPOLOGY_CACHE=0)
 *   mode 1  reuse within one evaluation (default)
 *   mode 2  reuse across evaluations (NQA_TOPOLOGY_CACHE=2): ONLY for hosts whose index tensors change solely through
 *           torch operations (new tensors or in-place torch ops, which bump the version counter); call
 *           nqa_torch_topology_invalidate() after any out-of-band rewrite.
 * NQA_TOPOLOGY_VERIFY=1 (environment, read once): every hit is checked against a device checksum of the index tensors and
 * rebuilt on a mismatch -- one host synchronisation per hit, for diagnosis.
 * nqa_torch_topology_cache_mode(m) sets the mode and returns the previous one (m outside 0..2: query only). */
int nqa_torch_topology_cache_mode(int mode);
void nqa_torch_topology_invalidate(void);
void nqa_torch_begin_evaluation(void);

/* node_linear tables of `key` (transposed != 0: the adjoint map): chunk records of 8 int32, instruction records of 4
 * int32 (include/nequip_amd.h, nqa_node_linear).  Returns (n_chunk_int32 << 16) | n_instr_int32, or -1; copies when the
 * capacities (in int32) suffice.  dims = {dim_in, dim_out, weight_stride}. */
int nqa_torch_linear_tables(const char* key, int transposed, int32_t* chunks, int32_t chunks_cap, int32_t* instr,
                            int32_t instr_cap, int64_t* dims);
/* gather indices that turn packed forward weights into the packed weights of the adjoint map; returns their number */
int64_t nqa_torch_linear_transpose_perm(const char* key, int64_t* out, int64_t cap);
/* gate column table (32-byte records; which = 0 forward, 1 backward); returns its size in bytes; dims = {dim_in, dim_out} */
int64_t nqa_torch_gate_table(const char* key, int which, uint8_t* out, int64_t cap, int64_t* dims);
/* the nqa_gate_block array (include/nequip_amd.h) that node_stage_fwd / _bwd derive from a gate key, as bytes; dims = {dim_in,
 * dim_out} of the gate; returns the byte count, -1 for a key whose gates and gated irreps do not correspond one to one */
int64_t nqa_torch_gate_blocks(const char* key, uint8_t* out, int64_t cap, int64_t* dims);
/* out7 = {dim_in1, dim_in2, dim_out, weight_numel, out_needs_zero, prefer_fused_bwd, fused_rows_ok} of a plan text */
int nqa_torch_plan_dims(const char* plan, int64_t* out7);

#ifdef __cplusplus
}
#endif

#endif /* NEQUIP_AMD_TORCH_H */
